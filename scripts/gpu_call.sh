#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r5_tests_full2.txt
cat gpurun_out/r5_tests_full2.txt
