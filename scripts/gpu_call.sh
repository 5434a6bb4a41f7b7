#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r5_host_mapped.txt
: > $O
timeout 1500 python -m pytest tests/test_search_gpu.py tests/test_random_ops_gpu.py tests/test_concurrency_gpu.py tests/test_persistence_gpu.py tests/test_centred_gpu.py tests/test_compressed_gpu.py tests/test_cpp_host.py -m gpu -x -q > /tmp/pt.log 2>&1
echo "pytest rc=$?" >> $O
grep -E "passed|failed|Error" /tmp/pt.log | tail -4 >> $O
MEMEX_HIP_SPIN=1 timeout 600 python scripts/gpu_small_corpus_latency.py 2>&1 | grep "n=" >> $O
echo "-- MEMEX_HIP_HOST_COPIES=1" >> $O
MEMEX_HIP_SPIN=1 MEMEX_HIP_HOST_COPIES=1 timeout 600 python scripts/gpu_small_corpus_latency.py 2>&1 | grep "n=" | head -8 >> $O
cat $O
