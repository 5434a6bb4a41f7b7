#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r5_scan8_16x16.txt
: > $O
timeout 900 python -m pytest tests/test_search_gpu.py tests/test_centred_gpu.py -m gpu -x -q 2>&1 | tail -15 >> $O
for rep in 1 2; do timeout 120 build_ub/scan8_new 10000000 384 1500 >> $O 2>&1; done
timeout 120 build_ub/scan8_new 10000000 768 800 >> $O 2>&1
timeout 400 python bench.py > gpurun_out/r5_bench_16x16.json 2> gpurun_out/r5_bench_16x16.err
tail -3 gpurun_out/r5_bench_16x16.err >> $O
cat gpurun_out/r5_bench_16x16.json >> $O
cat $O
