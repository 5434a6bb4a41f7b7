#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_centred_gpu.py tests/test_search_gpu.py tests/test_compressed_gpu.py tests/test_random_ops_gpu.py tests/test_cfg2_gpu.py -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r5_tests5.txt
cat gpurun_out/r5_tests5.txt
timeout 600 python scripts/gpu_enc_like.py 10000000 20 > gpurun_out/r5_enc_like_after.json 2> gpurun_out/r5_enc_like_after.err; tail -c 1500 gpurun_out/r5_enc_like_after.json; tail -5 gpurun_out/r5_enc_like_after.err
