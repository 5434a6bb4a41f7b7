#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_centred_gpu.py -m gpu -x -q > /tmp/pt.log 2>&1
echo "pytest rc=$?" > gpurun_out/r5_centred_wild.txt
tail -30 /tmp/pt.log >> gpurun_out/r5_centred_wild.txt
cat gpurun_out/r5_centred_wild.txt
