"""int8 against bf16 filter copy on rows that look more like sentence embeddings than i.i.d. Gaussians do (not a test):
a decaying spectrum (dimension i scaled by (i+1)^-p) plus a common mean direction (random pairs have cosine ~0.2-0.5),
queries drawn the same way.  MEMEX_HIP_FILTER=i8|bf16 pins the copy; unset = the library's choice (with demotion)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from memex_amd.index import FlatIndex

def rows(n, d, p, mean, gen):
    scale = (torch.arange(1, d + 1, device="cuda", dtype=torch.float32) ** (-p))
    mu = torch.zeros(d, device="cuda"); mu[0] = mean
    x = torch.randn((n, d), device="cuda", generator=gen) * scale + mu * scale.norm()
    return x

def run(n, d, p, mean, B=256, k=10, steps=8):
    idx = FlatIndex(d); idx.reserve(n)
    g = torch.Generator(device="cuda")
    for b0 in range(0, n, 1_000_000):
        g.manual_seed(b0 + 7); x = rows(min(1_000_000, n - b0), d, p, mean, g); torch.cuda.synchronize(); idx.add_device(x); del x
    g.manual_seed(99); q = rows(B, d, p, mean, g)
    ids = torch.zeros((B, k), dtype=torch.int64, device="cuda"); sc = torch.zeros((B, k), device="cuda"); di = torch.zeros((B, k), device="cuda"); nf = torch.zeros((B,), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for _ in range(3): idx.search_device(q, k, ids, sc, di, nf)
    idx.reset_stats(); idx.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(steps): idx.search_device(q, k, ids, sc, di, nf)
    dt = (time.perf_counter() - t0) / steps
    st = idx.stats()
    print(f"n={n} d={d} spectrum^-{p} mean={mean}: {dt*1e3:.3f} ms/step {B/dt:.0f} QPS kind={st.filter_kind} demotions={st.filter_demotions} retry={st.retry_queries} fallback={st.fallback_queries} "
          f"cand/q={st.candidates/max(1,st.queries):.0f} e1={st.approx_err_bound:.4f} top score {float(sc[:, 0].mean()):.3f} 10th {float(sc[:, -1].mean()):.3f}", flush=True)
    idx.close()

for p, mean in ((0.0, 0.0), (0.3, 0.0), (0.5, 0.0), (0.3, 0.3), (0.5, 0.6), (0.8, 0.3)):
    run(10_000_000, 384, p, mean)
