#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r5_cpp_host.txt
: > $O
timeout 900 python -m pytest tests/test_cpp_host.py -m gpu -x -q 2>&1 | tail -15 >> $O
cat $O
