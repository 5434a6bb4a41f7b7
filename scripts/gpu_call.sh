#!/bin/bash
# the GPU call of the moment (see scripts/README.md): soak of the random-operation and concurrency tests
mkdir -p gpurun_out
MEMEX_TEST_SOAK=4 timeout 2400 python -m pytest tests/test_random_ops_gpu.py -m gpu -q > /tmp/pt.log 2>&1
echo "pytest rc=$?" > gpurun_out/r5_soak_final.txt
grep -E "passed|failed|^E |FAILED" /tmp/pt.log | tail -12 >> gpurun_out/r5_soak_final.txt
cat gpurun_out/r5_soak_final.txt
