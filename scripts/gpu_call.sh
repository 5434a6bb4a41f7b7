#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_encoder_gpu.py tests/test_pretrained.py -m gpu -x -q > /tmp/pt.log 2>&1
echo "pytest rc=$?" > gpurun_out/r5_enc_tidy.txt
grep -E "passed|failed|Error" /tmp/pt.log | tail -3 >> gpurun_out/r5_enc_tidy.txt
cat gpurun_out/r5_enc_tidy.txt
