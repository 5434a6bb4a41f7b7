#!/bin/bash
# the GPU call of the moment (see scripts/README.md)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_centred_gpu.py tests/test_random_ops_gpu.py tests/test_search_gpu.py tests/test_persistence_gpu.py tests/test_compressed_gpu.py -m gpu -x -q > /tmp/pt.log 2>&1
echo "pytest rc=$?" > gpurun_out/r5_recentre.txt
grep -E "passed|failed|^E |FAILED" /tmp/pt.log | tail -10 >> gpurun_out/r5_recentre.txt
cat gpurun_out/r5_recentre.txt
