"""Single-query search latency on small collections (what a memex user has: thousands to a million segments), host API,
p50 / p99 over 300 calls.  usage: gpu_small_corpus_latency.py [dim]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from memex_amd.index import FlatIndex
d = int(sys.argv[1]) if len(sys.argv) > 1 else 384
rng = np.random.default_rng(0)
for n in (1000, 10_000, 100_000, 1_000_000):
    X = rng.standard_normal((n, d), dtype=np.float32)
    with FlatIndex(d) as idx:
        idx.add(X)
        for B in (1, 16):
            Q = rng.standard_normal((B, d), dtype=np.float32)
            for _ in range(20): idx.search(Q, 10)
            ts = []
            for _ in range(300):
                t0 = time.perf_counter(); idx.search(Q, 10); ts.append((time.perf_counter() - t0) * 1e3)
            st = idx.stats()
            print(f"n={n:8d} B={B:2d}: p50 {np.percentile(ts, 50):.3f} ms  p99 {np.percentile(ts, 99):.3f} ms  (kind {st.filter_kind}, fallbacks {st.fallback_queries})", flush=True)

# the same through the device-pointer entry point (no staging copies): what the host API's copies cost
import torch
from bench import SearchBuffers
for n in (1000, 100_000):
    X = rng.standard_normal((n, d), dtype=np.float32)
    with FlatIndex(d) as idx:
        idx.add(X)
        q = torch.randn((1, d), device="cuda")
        bufs = SearchBuffers(1, 10)
        for _ in range(20): idx.search_device(q, 10, bufs.ids, bufs.scores, bufs.dists, bufs.nf)
        ts = []
        for _ in range(300):
            t0 = time.perf_counter(); idx.search_device(q, 10, bufs.ids, bufs.scores, bufs.dists, bufs.nf); ts.append((time.perf_counter() - t0) * 1e3)
        print(f"n={n:8d} B= 1 device pointers: p50 {np.percentile(ts, 50):.3f} ms  p99 {np.percentile(ts, 99):.3f} ms", flush=True)
