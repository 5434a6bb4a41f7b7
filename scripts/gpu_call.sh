#!/bin/bash
# round 6: attention_x3_kernel with two sets of LDS tiles (one barrier per key block): parity tests, throughput
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_encoder_gpu.py tests/test_pipeline_native_gpu.py -m gpu -x -q > gpurun_out/r6z_tests.txt 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r6z_tests.txt
timeout 900 python scripts/gpu_encoder_precise.py 2>&1 | grep "chunks/s" | tee gpurun_out/r6z_precise.txt
