#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_concurrency_gpu.py -m gpu -x -q --durations=3 > /tmp/pt.log 2>&1
echo "pytest rc=$?" > gpurun_out/r5_concurrency.txt
tail -40 /tmp/pt.log >> gpurun_out/r5_concurrency.txt
cat gpurun_out/r5_concurrency.txt
