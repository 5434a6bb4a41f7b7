"""How a thread waits for the GPU (not a test): wall time per call and CPU time burnt meanwhile, for a search batch and for
encoder calls.  Default: the search sleeps for most of what recent batches took and then polls its completion word, the
encoder naps between event queries; MEMEX_HIP_SPIN=1 polls throughout (hipStreamSynchronize / hipEventSynchronize --
also on a blocking-sync event -- poll too: 100 % of a core)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from memex_amd.index import FlatIndex
n, d, B, k = 10_000_000, 384, 256, 10
idx = FlatIndex(d); idx.reserve(n)
g = torch.Generator(device="cuda")
for b0 in range(0, n, 1_000_000):
    g.manual_seed(b0); x = torch.randn((1_000_000, d), device="cuda", generator=g); torch.cuda.synchronize(); idx.add_device(x); del x
q = torch.randn((B, d), device="cuda", generator=g)
ids = torch.zeros((B, k), dtype=torch.int64, device="cuda"); sc = torch.zeros((B, k), device="cuda"); di = torch.zeros((B, k), device="cuda"); nf = torch.zeros((B,), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
for _ in range(5): idx.search_device(q, k, ids, sc, di, nf)
t0, c0 = time.perf_counter(), time.process_time()
for _ in range(400): idx.search_device(q, k, ids, sc, di, nf)
t1, c1 = time.perf_counter(), time.process_time()
print(f"SPIN={os.environ.get('MEMEX_HIP_SPIN')} search: {(t1-t0)/400*1e3:.4f} ms per batch, CPU busy {(c1-c0)/(t1-t0)*100:.0f} % of one core")

idx.close()
import numpy as np
from memex_amd import weights as W
from memex_amd.encoder import Encoder
cfg = W.ALL_MINILM_L6_V2
enc = Encoder(cfg, W.synthetic_weights(cfg, 3))
for Bc, S, reps in ((2048, 512, 20), (1, 16, 400)):
    ids = torch.randint(1000, cfg.vocab, (Bc, S), dtype=torch.int32, device="cuda"); lens = torch.full((Bc,), S, dtype=torch.int32, device="cuda")
    out = torch.zeros((Bc, cfg.hidden), device="cuda")
    for _ in range(3): enc.encode_device(ids, lens, out)
    t0, c0 = time.perf_counter(), time.process_time()
    for _ in range(reps): enc.encode_device(ids, lens, out)
    t1, c1 = time.perf_counter(), time.process_time()
    print(f"SPIN={os.environ.get('MEMEX_HIP_SPIN')} encode {Bc} x {S}: {(t1-t0)/reps*1e3:.3f} ms per call, CPU busy {(c1-c0)/(t1-t0)*100:.0f} % of one core")
enc.close()
