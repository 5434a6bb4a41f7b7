#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_concurrency_gpu.py -m gpu -x -q --durations=3 -k "96" > /tmp/pt.log 2>&1
echo "pytest rc=$?" > gpurun_out/r5_concurrency_sharded.txt
tail -40 /tmp/pt.log >> gpurun_out/r5_concurrency_sharded.txt
cat gpurun_out/r5_concurrency_sharded.txt
