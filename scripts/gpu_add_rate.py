"""Append rate from HBM by filter-copy kind (not a test): ingest + filter-copy construction per 1M rows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from memex_amd.index import FlatIndex
for d in (384, 768, 1024):
    for kind in ("i8", "bf16"):
        idx = FlatIndex(d); idx.set_filter_copy(kind); idx.reserve(4_000_000)
        g = torch.Generator(device="cuda"); g.manual_seed(1)
        x = torch.randn((1_000_000, d), device="cuda", generator=g); torch.cuda.synchronize()
        idx.add_device(x)
        t0 = time.perf_counter()
        for _ in range(3): idx.add_device(x)
        dt = (time.perf_counter() - t0) / 3
        print(f"d={d} {kind}: add_device of 1M rows {dt*1e3:.2f} ms ({1e6/dt/1e6:.0f}M rows/s, {1e6*d*4/dt/1e9:.0f} GB/s of f32 in)", flush=True)
        idx.close()
