#!/bin/bash
# round 4, call 3: pgemm_kernel inside the encoder: parity tests, then in-situ A/B of chunks/s
mkdir -p gpurun_out
out=gpurun_out/r4_enc_ab.txt
: > $out
timeout 900 python -m pytest tests/test_encoder_gpu.py -x -q -m gpu -k "pgemm or vs_oracle or fused_layer" 2>&1 | tail -15 >> $out
for pg in 0 1 0 1; do
  MEMEX_HIP_PGEMM=$pg timeout 300 python scripts/r4_enc_ab.py both 6 2>&1 | grep -v amdgpu.ids >> $out
done
cat $out
