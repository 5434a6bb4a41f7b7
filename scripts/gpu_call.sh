#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_persistence_gpu.py -m gpu -x -q > /tmp/pt.log 2>&1
echo "pytest rc=$?" > gpurun_out/r5_persist.txt
tail -40 /tmp/pt.log >> gpurun_out/r5_persist.txt
cat gpurun_out/r5_persist.txt
