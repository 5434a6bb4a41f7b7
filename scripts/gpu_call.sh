#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > /tmp/pt.log 2>&1
echo "pytest rc=$?" > gpurun_out/r5_full_check2.txt
grep -E "passed|failed|error" /tmp/pt.log | tail -5 >> gpurun_out/r5_full_check2.txt
cat gpurun_out/r5_full_check2.txt
