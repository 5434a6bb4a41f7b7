#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_pipeline_native_gpu.py tests/test_pretrained.py -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r5_tests4.txt
cat gpurun_out/r5_tests4.txt
timeout 300 python scripts/gpu_query_latency.py > gpurun_out/r5_query_latency.txt 2>&1
cat gpurun_out/r5_query_latency.txt
bash scripts/small_pass_trace.sh > /dev/null 2>&1; cp gpurun_out/small_pass_trace.txt gpurun_out/r5_small_pass_trace.txt; head -12 gpurun_out/r5_small_pass_trace.txt; tail -3 gpurun_out/r5_small_pass_trace.txt
