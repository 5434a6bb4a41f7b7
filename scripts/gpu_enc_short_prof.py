"""Kernel-trace target: all-MiniLM-L12-v2 shape at its own window (max_seq_length 128), full passes; argv: B S [ragged]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from memex_amd.encoder import Encoder
from memex_amd import weights as W
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
ragged = len(sys.argv) > 3 and sys.argv[3] == "ragged"   # argv[4]: l12 | roberta | bge
cfg = {"l12": W.ALL_MINILM_L12_V2, "roberta": W.ALL_DISTILROBERTA_V1, "bge": W.BGE_BASE_EN}[sys.argv[4] if len(sys.argv) > 4 else "l12"]
enc = Encoder(cfg, W.synthetic_weights(cfg, 0))
g = torch.Generator(device="cuda"); g.manual_seed(1)
ids = torch.randint(1000, cfg.vocab, (B, S), device="cuda", dtype=torch.int32, generator=g)
lens = (torch.randint(S // 4, S + 1, (B,), device="cuda", dtype=torch.int32, generator=g) if ragged
        else torch.full((B,), S, device="cuda", dtype=torch.int32))
out = torch.zeros((B, cfg.hidden), device="cuda")
torch.cuda.synchronize()
for _ in range(3): enc.encode_device(ids, lens, out)
torch.cuda.synchronize()
