#!/bin/bash
# round 6: the checkpoint-like-weights test with its printed errors (MX_PREC_MIXED with P as one bf16 value in P.V)
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_encoder_gpu.py -m gpu -x -q -s -k "checkpoint_like" 2>&1 | grep "checkpoint-like weights\|passed\|failed" | tee gpurun_out/r6t_checkpoint_like.txt
