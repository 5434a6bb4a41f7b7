#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_search_gpu.py tests/test_sharded_gpu.py tests/test_random_ops_gpu.py tests/test_persistence_gpu.py -q -m gpu -rf 2>&1 | grep -v "^$" | grep -v "^E    \|^    " | tail -80 > gpurun_out/r4_gputests2.txt
cat gpurun_out/r4_gputests2.txt
