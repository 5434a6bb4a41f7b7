#!/bin/bash
# round 4, call 1: GEMM microbenchmark on one bge-base layer's shapes (131072 tokens): gemm_kernel configurations and
# their ablations (no DMA in the loop / no epilogue / neither), pgemm_kernel and its ablations (+ no stagger / no setprio)
mkdir -p gpurun_out
out=gpurun_out/r4_gemm_ub.txt
: > $out
for b in gemm_ub gemm_ub_a1 gemm_ub_a2 gemm_ub_a3 gemm_ub_p4 gemm_ub_p8; do
  echo "=== $b" >> $out
  timeout 240 build_ub/$b 131072 768 3072 50 >> $out 2>&1
  echo "rc=$?" >> $out
done
echo "=== gemm_ub MiniLM shapes" >> $out
timeout 240 build_ub/gemm_ub 131072 384 1536 50 >> $out 2>&1
echo "rc=$?" >> $out
cat $out
