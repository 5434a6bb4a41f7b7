"""Encoder throughput A/B inside one process per setting (not a test): chunks/s of 512-token chunks for the models of
BASELINE configs[4] / configs[3], with the environment switches given on the command line as KEY=VALUE.
usage: python scripts/r4_enc_ab.py [minilm|bge|both] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from memex_amd.encoder import Encoder
from memex_amd import weights as W


def run(name, cfg, B, S, reps):
    w = W.synthetic_weights(cfg, 0)
    enc = Encoder(cfg, w)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    ids = torch.randint(1000, cfg.vocab, (B, S), device="cuda", dtype=torch.int32, generator=g)
    lens = torch.full((B,), S, device="cuda", dtype=torch.int32)
    out = torch.zeros((B, cfg.hidden), device="cuda")
    torch.cuda.synchronize()
    enc.encode_device(ids, lens, out); enc.encode_device(ids, lens, out)
    enc.reset_stats(); enc.set_profiling(True)
    for _ in range(reps): enc.encode_device(ids, lens, out)
    st = enc.stats()
    sw = {k: v for k, v in os.environ.items() if k.startswith("MEMEX_HIP_") and k != "MEMEX_HIP_SPIN"}
    print(f"{name} {sw}: {st.sequences/(st.gpu_ms/1e3):.0f} chunks/s, {st.flops/(st.gpu_ms/1e3)/1e12:.1f} TFLOP/s "
          f"= {st.flops/(st.gpu_ms/1e3)/2.5e15:.3f} of bf16 MFMA peak", flush=True)
    enc.close()


which = sys.argv[1] if len(sys.argv) > 1 else "both"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
if which in ("bge", "both"): run("bge-base", W.BGE_BASE_EN, 1024, 512, reps)
if which in ("minilm", "both"): run("minilm-l6", W.ALL_MINILM_L6_V2, 2048, 512, reps)
