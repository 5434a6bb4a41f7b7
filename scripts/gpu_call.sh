#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r5_small_chain.txt
: > $O
timeout 1500 python -m pytest tests/test_encoder_gpu.py tests/test_pretrained.py tests/test_pipeline_native_gpu.py -m gpu -x -q > /tmp/pt.log 2>&1
echo "pytest rc=$?" >> $O
grep -E "passed|failed|Error|assert" /tmp/pt.log | tail -6 >> $O
timeout 300 python scripts/gpu_query_latency.py l12,l6 1x16,1x128,8x32 >> $O 2>&1
echo "-- MEMEX_HIP_SMALL_CHAIN=0" >> $O
MEMEX_HIP_SMALL_CHAIN=0 timeout 300 python scripts/gpu_query_latency.py l12,l6 1x16,1x128,8x32 >> $O 2>&1
cat $O
