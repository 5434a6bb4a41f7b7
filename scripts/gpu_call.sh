#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_pretrained.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r5_tests8.txt
cat gpurun_out/r5_tests8.txt
