import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from memex_amd.encoder import Encoder
from memex_amd import weights as W
import dataclasses
cfg = W.ALL_MINILM_L6_V2 if len(sys.argv) < 2 or sys.argv[1] == "l6" else W.BGE_BASE_EN
if len(sys.argv) > 2: cfg = dataclasses.replace(cfg, precision=sys.argv[2])   # "bf16x3"
B = (2048 if cfg.hidden == 384 else 1024) // (4 if cfg.precision in ("bf16x3", "mixed", "mixed1") else 1)
enc = Encoder(cfg, W.synthetic_weights(cfg, 0))
ids = torch.randint(1000, cfg.vocab, (B, 512), device="cuda", dtype=torch.int32)
lens = torch.full((B,), 512, device="cuda", dtype=torch.int32)
out = torch.zeros((B, cfg.hidden), device="cuda")
torch.cuda.synchronize()
for _ in range(3): enc.encode_device(ids, lens, out)
