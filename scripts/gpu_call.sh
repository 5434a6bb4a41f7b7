#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r5_cone_seq.txt
: > $O
timeout 600 python -m pytest tests/test_random_ops_gpu.py -m gpu -q -k "7 or 8 or 9" 2>&1 | tail -60 >> $O
cat $O
