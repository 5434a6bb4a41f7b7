#!/bin/bash
# encoder tests with the final attention rule (<= 128 tokens: attention_short_kernel), pipeline tests, throughput probe
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_encoder_gpu.py tests/test_pipeline_native_gpu.py tests/test_cfg2_gpu.py tests/test_pretrained.py -m gpu -q 2>&1 | grep -E "passed|failed|rror|FAILED" | tail -8 ) > gpurun_out/r5z_tests.txt
timeout 600 python scripts/gpu_encoder_perf.py short 2>&1 | grep chunks > gpurun_out/r5z_perf.txt
cat gpurun_out/r5z_tests.txt gpurun_out/r5z_perf.txt
