"""A/B of the fused layer-tail kernel against the GEMM-by-GEMM path (not a test): output agreement + throughput."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from memex_amd.encoder import Encoder
from memex_amd import weights as W

def run(cfg, B, S, unfused, reps=5):
    os.environ["MEMEX_HIP_DEBUG"] = "unfused_tail=1" if unfused else ""
    enc = Encoder(cfg, W.synthetic_weights(cfg, 0))
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    ids = torch.randint(1000, cfg.vocab, (B, S), device="cuda", dtype=torch.int32, generator=g)
    lens = torch.full((B,), S, device="cuda", dtype=torch.int32)
    out = torch.zeros((B, cfg.hidden), device="cuda")
    enc.encode_device(ids, lens, out)
    enc.reset_stats(); enc.set_profiling(True)
    for _ in range(reps): enc.encode_device(ids, lens, out)
    st = enc.stats()
    print(f"L{cfg.layers} B={B} S={S} unfused={unfused}: {st.sequences/(st.gpu_ms/1e3):.0f} chunks/s, "
          f"{st.flops/(st.gpu_ms/1e3)/1e12:.1f} TFLOP/s ({st.flops/(st.gpu_ms/1e3)/2.5e15*100:.1f}%)", flush=True)
    enc.close()
    return out.clone()

for cfg, B, S in ((W.ALL_MINILM_L6_V2, 2048, 512), (W.ALL_MINILM_L6_V2, 64, 128), (W.ALL_MINILM_L12_V2, 512, 256)):
    a = run(cfg, B, S, True)
    b = run(cfg, B, S, False)
    cos = torch.nn.functional.cosine_similarity(a, b, dim=1)
    print(f"  fused vs unfused: min cosine {cos.min().item():.6f}, max |diff| {(a - b).abs().max().item():.2e}", flush=True)
