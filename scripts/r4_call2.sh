#!/bin/bash
# round 4, call 2: pgemm_kernel after the B-hi fix: bit-identity, start-up skew sweep, fewer CUs (per-CU or chip-wide store limit?)
mkdir -p gpurun_out
out=gpurun_out/r4_gemm_ub2.txt
: > $out
for sk in 0 1 2 4 8; do
  echo "=== skew $sk" >> $out
  MEMEX_HIP_PGEMM_SKEW=$sk timeout 240 build_ub/gemm_ub 131072 768 3072 50 2>&1 | grep "pgemm\|rc=" >> $out
done
echo "=== 64 CUs, skew 0 (time x 4 = per-CU-limited; less = chip-limited)" >> $out
MEMEX_HIP_PGEMM_SKEW=0 MEMEX_HIP_PGEMM_CUS=64 timeout 240 build_ub/gemm_ub 131072 768 3072 20 2>&1 | grep "pgemm" >> $out
cat $out
