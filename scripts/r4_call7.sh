#!/bin/bash
# round 4, call 7: pgemm_kernel: blocked tile order (MEMEX_HIP_PGEMM_GN) and the relaxed wait behind a tile boundary
# (gemm_ub_p16 = the old vmcnt(6) there), alone on the chip; then the encoder in situ
mkdir -p gpurun_out
out=gpurun_out/r4_gemm_ub3.txt
: > $out
for gn in 0 2 3 4 6; do
  echo "=== vmcnt(22) behind a boundary, GN=$gn" >> $out
  MEMEX_HIP_PGEMM_GN=$gn timeout 240 build_ub/gemm_ub 131072 768 3072 50 2>&1 | grep "pgemm" >> $out
done
echo "=== vmcnt(6) behind a boundary (MX_PGEMM_ABLATE=16), GN=0" >> $out
MEMEX_HIP_PGEMM_GN=0 timeout 240 build_ub/gemm_ub_p16 131072 768 3072 50 2>&1 | grep "pgemm" >> $out
echo "=== MiniLM shapes, GN=4" >> $out
MEMEX_HIP_PGEMM_GN=4 timeout 240 build_ub/gemm_ub 131072 384 1536 50 2>&1 | grep "pgemm" >> $out
for gn in 0 4; do for rep in 1 2; do
  MEMEX_HIP_PGEMM_GN=$gn timeout 300 python scripts/r4_enc_ab.py both 6 2>&1 | grep -v amdgpu.ids >> $out
done; done
timeout 900 python -m pytest tests/test_encoder_gpu.py -x -q -m gpu -k "pgemm or checkpoint" 2>&1 | tail -3 >> $out
cat $out
