#!/bin/bash
# the current GPU call (overwritten per call; recipes worth keeping are described in scripts/README.md)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r5_tests1.txt
cat gpurun_out/r5_tests1.txt
timeout 600 python scripts/gpu_encoder_precise.py > gpurun_out/r5_precise_perf.txt 2>&1
cat gpurun_out/r5_precise_perf.txt
timeout 900 python bench.py > gpurun_out/r5_bench1.json 2> gpurun_out/r5_bench1.err
tail -c 3000 gpurun_out/r5_bench1.json
