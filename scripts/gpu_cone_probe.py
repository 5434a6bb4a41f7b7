"""How dense is the neighbourhood of a query in the enc_like corpus (rows expanded from random-weight encoder outputs)?
Counts of rows within a margin of the 10th-best cosine, before and after removing the corpus mean direction."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from memex_amd import weights as W
from memex_amd.encoder import Encoder
cfg = W.ALL_MINILM_L6_V2
n_seg, rows, S, B = 100_000, int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, 128, 64
g = torch.Generator(device="cuda"); g.manual_seed(2025)
ids = torch.randint(1000, cfg.vocab, (n_seg, S), device="cuda", dtype=torch.int32, generator=g)
lens = torch.randint(16, S + 1, (n_seg,), device="cuda", dtype=torch.int32, generator=g)
qids = torch.randint(1000, cfg.vocab, (256, 32), device="cuda", dtype=torch.int32, generator=g)
qlens = torch.randint(4, 33, (256,), device="cuda", dtype=torch.int32, generator=g)
vec = torch.zeros((n_seg, cfg.hidden), device="cuda"); q = torch.zeros((256, cfg.hidden), device="cuda")
enc = Encoder(cfg, W.pack_weights(W.synthetic_weights(cfg, 0), cfg))
for b0 in range(0, n_seg, 16384): enc.encode_device(ids[b0:b0 + 16384], lens[b0:b0 + 16384], vec[b0:b0 + 16384])
enc.encode_device(qids, qlens, q); enc.close()
q = q[:B]
m = vec.mean(0); m = m / m.norm()
print("mean direction: a_c = c.m mean %.4f min %.4f; |r_c| mean %.4f max %.4f; a_q mean %.4f |r_q| mean %.4f" % (
    float((vec @ m).mean()), float((vec @ m).min()), float((1 - (vec @ m) ** 2).clamp_min(0).sqrt().mean()),
    float((1 - (vec @ m) ** 2).clamp_min(0).sqrt().max()), float((q @ m).mean()), float((1 - (q @ m) ** 2).clamp_min(0).sqrt().mean())))
margins = [2e-5, 1e-4, 2.5e-4, 5e-4, 1e-3, 2.1e-3, 4.3e-3, 8.6e-3]
cnt = torch.zeros((B, len(margins)), device="cuda", dtype=torch.int64)
top = torch.full((B, 10), -2.0, device="cuda")
blocks = []
for b0 in range(0, rows, 1_000_000):
    nb = min(1_000_000, rows - b0)
    src = torch.randint(0, n_seg, (nb,), device="cuda", generator=g)
    xb = vec[src] + (0.1 / cfg.hidden ** 0.5) * torch.randn((nb, cfg.hidden), device="cuda", generator=g)
    xb = xb / xb.norm(dim=1, keepdim=True)
    blocks.append(xb.half())
    cs = q @ xb.T
    top = torch.cat([top, cs.topk(10, dim=1).values], 1).topk(10, dim=1).values
k10 = top[:, 9:10]
mu = torch.zeros(B, device="cuda"); sq = torch.zeros(B, device="cuda")
for xb in blocks:
    cs = q @ xb.float().T
    mu += cs.sum(1); sq += (cs * cs).sum(1)
    for j, mg in enumerate(margins): cnt[:, j] += (cs >= k10 - mg).sum(1)
mu /= rows; sd = (sq / rows - mu * mu).sqrt()
print("rows %d: cos mean %.4f sd %.5f; 10th best mean %.4f (= mean + %.2f sd)" % (rows, float(mu.mean()), float(sd.mean()), float(k10.mean()), float(((k10[:, 0] - mu) / sd).mean())))
for j, mg in enumerate(margins):
    c = cnt[:, j].float()
    print("rows within %.1e of the 10th best: median %8d  p90 %8d  max %8d   queries with > 16384: %d / %d" % (
        mg, int(c.median()), int(c.quantile(0.9)), int(c.max()), int((c > 16384).sum()), B))
