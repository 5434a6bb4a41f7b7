#!/bin/bash
# round 6: MX_PREC_MIXED1 (the MLP on one fp16 product): encoder parity tests with printed errors, throughput
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_encoder_gpu.py tests/test_abi.py -m gpu -x -q -s 2>&1 | grep -E "checkpoint-like weights|pgemm_kernel, hidden|passed|failed|Error|error" | tee gpurun_out/r6aa_tests.txt | tail -24
timeout 900 python scripts/gpu_encoder_precise.py 2>&1 | grep "chunks/s" | tee gpurun_out/r6aa_precise.txt
