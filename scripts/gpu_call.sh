#!/bin/bash
# attention: head pairs chosen per pass (longest sequence <= 256 tokens, >= 1024 items) -- tests, throughput, query latency
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_encoder_gpu.py tests/test_pipeline_native_gpu.py tests/test_cfg2_gpu.py tests/test_pretrained.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -6 ) > gpurun_out/r5p_tests.txt
timeout 600 python scripts/gpu_encoder_perf.py short 2>&1 | grep chunks > gpurun_out/r5p_perf.txt
timeout 300 python scripts/gpu_query_latency.py > gpurun_out/r5p_query_latency.txt 2>&1
cat gpurun_out/r5p_tests.txt gpurun_out/r5p_perf.txt; tail -12 gpurun_out/r5p_query_latency.txt
