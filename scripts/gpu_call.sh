#!/bin/bash
# r6l: the mixed mode (MX_PREC_MIXED): parity tests, then throughput next to bf16x3
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_encoder_gpu.py -m gpu -x -q -k "vs_oracle or checkpoint_like or split_operand" -s 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r6l_tests.txt
cat gpurun_out/r6l_tests.txt
python - <<'P' 2>&1 | tee gpurun_out/r6l_mixed_perf.txt
import dataclasses, time, numpy as np, torch
from memex_amd import weights as W
from memex_amd.encoder import Encoder
for name, base, B in (("all-MiniLM-L6-v2", W.ALL_MINILM_L6_V2, 256), ("bge-base-en", W.BGE_BASE_EN, 256)):
    for prec in ("bf16x3", "mixed", "bf16"):
        cfg = dataclasses.replace(base, precision=prec)
        enc = Encoder(cfg, W.pack_weights(W.synthetic_weights(cfg, 0), cfg))
        g = torch.Generator(device="cuda"); g.manual_seed(3)
        ids = torch.randint(1000, cfg.vocab, (B, 512), device="cuda", dtype=torch.int32, generator=g)
        lens = torch.full((B,), 512, device="cuda", dtype=torch.int32)
        emb = torch.zeros((B, cfg.hidden), device="cuda")
        for _ in range(3): enc.encode_device(ids, lens, emb)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 24 if prec != "bf16" else 60
        for _ in range(n): enc.encode_device(ids, lens, emb)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{name:18s} {prec:7s} {n * B / dt:9.0f} chunks/s")
        enc.close()
P
