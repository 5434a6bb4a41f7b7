"""Encoder throughput probe (not a test): chunks/s and achieved TFLOP/s on synthetic 512-token chunks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from memex_amd.encoder import Encoder
from memex_amd import weights as W

def run(cfg, B, S, reps=5, ragged=False):
    w = W.synthetic_weights(cfg, 0)
    enc = Encoder(cfg, w)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    ids = torch.randint(1000, cfg.vocab, (B, S), device="cuda", dtype=torch.int32, generator=g)
    lens = (torch.randint(min(64, S // 4), S + 1, (B,), device="cuda", dtype=torch.int32, generator=g) if ragged
            else torch.full((B,), S, device="cuda", dtype=torch.int32))
    out = torch.zeros((B, cfg.hidden), device="cuda")
    torch.cuda.synchronize()
    enc.encode_device(ids, lens, out)
    enc.reset_stats(); enc.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(reps): enc.encode_device(ids, lens, out)
    dt = time.perf_counter() - t0
    st = enc.stats()
    print(f"L{cfg.layers} H{cfg.hidden} B={B} S={S} ragged={ragged}: {B*reps/dt:.0f} chunks/s wall, "
          f"{st.sequences/(st.gpu_ms/1e3):.0f} chunks/s gpu, {st.flops/(st.gpu_ms/1e3)/1e12:.1f} TFLOP/s "
          f"({st.flops/(st.gpu_ms/1e3)/2.5e15*100:.1f}% of 2.5 PF), tokens={st.tokens}")
    enc.close()

run(W.ALL_MINILM_L6_V2, 2048, 512)
run(W.ALL_MINILM_L6_V2, 2048, 512, ragged=True)
run(W.ALL_MINILM_L6_V2, 4096, 256)
run(W.BGE_BASE_EN, 1024, 512, reps=3)
run(W.ALL_MINILM_L12_V2, 64, 128)
if len(sys.argv) > 1 and sys.argv[1] == "short":   # the reference's default model at its own window: max_seq_length 128
    run(W.ALL_MINILM_L12_V2, 8192, 128, reps=3)
    run(W.ALL_MINILM_L12_V2, 8192, 128, reps=3, ragged=True)
    run(W.ALL_MINILM_L6_V2, 16384, 64, reps=3)
