#!/bin/bash
# A/B on one box: computed rows = round_up(rows, 256) (default) against round_up(rows + 32, 256) (MEMEX_HIP_PAD_TILE=1)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in 1 2; do for v in 0 1; do echo "== MEMEX_HIP_PAD_TILE=$v (run $rep)"; MEMEX_HIP_PAD_TILE=$v timeout 300 python scripts/gpu_encoder_perf.py 2>&1 | grep chunks; done; done > gpurun_out/r5f_pad_tile_ab.txt
cat gpurun_out/r5f_pad_tile_ab.txt
