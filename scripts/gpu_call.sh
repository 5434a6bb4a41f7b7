#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_pipeline_native_gpu.py tests/test_pretrained.py tests/test_cfg2_gpu.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r5_tests7.txt
cat gpurun_out/r5_tests7.txt
timeout 300 python scripts/gpu_query_latency.py > gpurun_out/r5_query_latency2.txt 2>&1
cat gpurun_out/r5_query_latency2.txt
bash scripts/small_pass_trace.sh > /dev/null 2>&1; cp gpurun_out/small_pass_trace.txt gpurun_out/r5_small_pass_trace2.txt; head -4 gpurun_out/r5_small_pass_trace2.txt; tail -4 gpurun_out/r5_small_pass_trace2.txt
