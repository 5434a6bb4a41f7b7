#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_pipeline_native_gpu.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r5_tests10.txt
cat gpurun_out/r5_tests10.txt
