"""Search throughput on other shapes (not a test): d=768 (cfg 4 per-GPU shard), B=1, k=100."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from memex_amd.index import FlatIndex
def run(n, d, B, k, steps=10):
    idx = FlatIndex(d); idx.reserve(n)
    g = torch.Generator(device="cuda")
    for b0 in range(0, n, 1_000_000):
        g.manual_seed(b0)
        x = torch.randn((min(1_000_000, n - b0), d), device="cuda", generator=g); torch.cuda.synchronize()
        idx.add_device(x); del x
    q = torch.randn((B, d), device="cuda", generator=g)
    ids = torch.zeros((B, k), dtype=torch.int64, device="cuda"); sc = torch.zeros((B, k), device="cuda"); di = torch.zeros((B, k), device="cuda"); nf = torch.zeros((B,), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for _ in range(2): idx.search_device(q, k, ids, sc, di, nf)
    idx.reset_stats(); idx.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(steps): idx.search_device(q, k, ids, sc, di, nf)
    dt = (time.perf_counter() - t0) / steps
    st = idx.stats()
    print(f"n={n} d={d} B={B} k={k}: {dt*1e3:.3f} ms/step {B/dt:.0f} QPS scan {st.scan_bytes/max(st.scan_ms,1e-9)/1e6:.0f} GB/s ({st.scan_ms/max(1,st.scan_launches):.3f} ms x {st.scan_launches}) fallback={st.fallback_queries} retry={st.retry_queries} cand/q={st.candidates/max(1,st.queries):.0f} e1={st.approx_err_bound:.4f} filter={st.filter_copy_bytes/max(1,n)/d:.1f} B/elem")
    idx.close()
if len(sys.argv) > 1 and sys.argv[1] == "small":     # small batches: waves without live queries skip their MFMAs
    for B in (1, 8, 32, 33, 64, 128, 256):
        run(10_000_000, 384, B, 10, 10)
    run(10_000_000, 768, 1, 10, 10)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "bulk":      # more than 256 queries per call: 512 per pass (int8 copy, <= 512 dims)
    for B in (256, 512, 1024, 2048):
        run(10_000_000, 384, B, 10, 6)
    run(10_000_000, 256, 1024, 10, 6)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "topk":      # larger k: the sample grows with k
    for kk in (10, 30, 100, 256):
        run(10_000_000, 384, 256, kk, 6)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "dims":      # MEMEX_HIP_FILTER=i8|bf16: which filter copy pays at which width
    for n, d in ((10_000_000, 384), (10_000_000, 512), (10_000_000, 768), (4_000_000, 1024), (4_000_000, 1536), (10_000_000, 256), (10_000_000, 128)):
        run(n, d, 256, 10, 6)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "wide":      # wide rows: scan16w_kernel, 128 queries per pass
    run(4_000_000, 1024, 128, 10, 6)
    run(4_000_000, 1024, 256, 10, 6)
    run(4_000_000, 1536, 128, 10, 6)
    run(4_000_000, 1536, 256, 10, 6)
    run(4_000_000, 1280, 128, 10, 6)
    run(4_000_000, 1024, 1, 10, 6)
    sys.exit(0)
run(10_000_000, 768, 256, 10, 6)
run(10_000_000, 384, 1, 10)
run(10_000_000, 384, 32, 10)
run(10_000_000, 384, 256, 100)
run(1_000_000, 384, 256, 10)
run(100_000, 384, 256, 10, 30)
