"""Per-step wall time of the headline search loop (10M x 384, B = 256, k = 10): percentiles and the slow steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from memex_amd.index import FlatIndex
n, d, B, k = int(os.environ.get("ROWS", "10000000")), 384, 256, 10
idx = FlatIndex(d); idx.reserve(n)
for b0 in range(0, n, 1_000_000):
    idx.add_device(bench.gaussian_rows(min(1_000_000, n - b0), d, 1234 + b0 // 1_000_000, "cuda"))
q = bench.gaussian_rows(B, d, 4321, "cuda")
ids = torch.zeros((B, k), dtype=torch.int64, device="cuda"); sc = torch.zeros((B, k), device="cuda"); di = torch.zeros((B, k), device="cuda"); nf = torch.zeros((B,), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
for _ in range(10): idx.search_device(q, k, ids, sc, di, nf)
if os.environ.get("PROF"): idx.reset_stats(); idx.set_profiling(True)
ts = []
for i in range(int(os.environ.get("STEPS", "300"))):
    t0 = time.perf_counter(); idx.search_device(q, k, ids, sc, di, nf); ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
print("ms per step: median %.3f  p10 %.3f  p90 %.3f  p99 %.3f  max %.3f  mean %.3f" % (np.median(ts), np.percentile(ts, 10), np.percentile(ts, 90), np.percentile(ts, 99), ts.max(), ts.mean()))
slow = np.nonzero(ts > 1.15 * np.median(ts))[0]
print("slow steps (>1.15 x median):", len(slow), slow[:30].tolist(), np.round(ts[slow[:30]], 3).tolist())
idx.close()
