#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r5_splitk_qkv.txt
: > $O
timeout 1500 python -m pytest tests/test_encoder_gpu.py tests/test_pretrained.py -m gpu -x -q -s > /tmp/pt.log 2>&1
echo "pytest rc=$?" >> $O
grep -E "passed|failed|Error|split-k small" /tmp/pt.log | tail -8 >> $O
timeout 300 python scripts/gpu_query_latency.py bge,roberta 1x16,8x32,3x128,8x128,24x128 >> $O 2>&1
cat $O
