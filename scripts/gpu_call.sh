#!/bin/bash
# bf16 MiniLM: the V projection (EPI_VT, N = 384) on pgemm_kernel with a half-valid second column tile (MEMEX_HIP_PGEMM_PART_VT=1)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( MEMEX_HIP_PGEMM_PART_VT=1 timeout 900 python -m pytest tests/test_encoder_gpu.py -m gpu -x -q -k "pgemm_kernel or full_passes or vs_oracle" 2>&1 | grep -E "passed|failed|rror|assert" | tail -6 ) > gpurun_out/r5l_tests.txt
for v in 1 0 1 0; do echo "== MEMEX_HIP_PGEMM_PART_VT=$v"; MEMEX_HIP_PGEMM_PART_VT=$v timeout 300 python scripts/gpu_encoder_perf.py 2>&1 | grep -E "B=2048 S=512 ragged=False" ; done > gpurun_out/r5l_part_vt_ab.txt
cat gpurun_out/r5l_tests.txt gpurun_out/r5l_part_vt_ab.txt
