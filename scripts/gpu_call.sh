#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_encoder_gpu.py -m gpu -q -s -k "over_weight_seeds" 2>&1 | grep "checkpoint-like weights\|passed\|failed\|assert" | tee gpurun_out/r6ff_seeds.txt
