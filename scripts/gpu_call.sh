#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r5_soak.txt
: > $O
MEMEX_TEST_SOAK=5 timeout 1500 python -m pytest tests/test_random_ops_gpu.py -m gpu -x -q 2>&1 | tail -8 >> $O
cat $O
