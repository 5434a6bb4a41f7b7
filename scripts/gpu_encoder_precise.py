"""MX_PREC_BF16X3 / MX_PREC_MIXED throughput probe (not a test): chunks/s of the split-operand modes next to the bf16 default."""
import dataclasses, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from memex_amd.encoder import Encoder
from memex_amd import weights as W

def run(cfg, B, S, reps=3):
    w = W.pack_weights(W.synthetic_weights(cfg, 0), cfg)
    for prec in ("bf16", "bf16x3", "mixed", "mixed1"):
        c = dataclasses.replace(cfg, precision=prec)
        enc = Encoder(c, w)
        g = torch.Generator(device="cuda"); g.manual_seed(1)
        ids = torch.randint(1000, cfg.vocab, (B, S), device="cuda", dtype=torch.int32, generator=g)
        lens = torch.full((B,), S, device="cuda", dtype=torch.int32)
        out = torch.zeros((B, cfg.hidden), device="cuda")
        enc.encode_device(ids, lens, out)
        enc.reset_stats(); enc.set_profiling(True)
        for _ in range(reps): enc.encode_device(ids, lens, out)
        st = enc.stats()
        print(f"L{cfg.layers} H{cfg.hidden} B={B} S={S} {prec}: {st.sequences/(st.gpu_ms/1e3):.0f} chunks/s gpu, "
              f"{st.flops/(st.gpu_ms/1e3)/1e12:.1f} algorithmic TFLOP/s", flush=True)
        enc.close()

run(W.ALL_MINILM_L6_V2, 1024, 512)
run(W.BGE_BASE_EN, 256, 512)
run(W.ALL_MINILM_L12_V2, 1, 32, reps=20)
