#!/bin/bash
# token_map: ranks of the attention work list counted by the whole grid -- encoder tests, throughput at short windows
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_encoder_gpu.py tests/test_pipeline_native_gpu.py tests/test_cfg2_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -6 ) > gpurun_out/r5q_tests.txt
timeout 600 python scripts/gpu_encoder_perf.py short 2>&1 | grep chunks > gpurun_out/r5q_perf.txt
cat gpurun_out/r5q_tests.txt gpurun_out/r5q_perf.txt
