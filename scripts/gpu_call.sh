#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r5_pgemm_shape_probe.txt
: > $O
for rep in 1 2; do for a in 0 32; do timeout 200 build_ub/gemm_ub_p$a 131072 768 3072 50 2>&1 | grep -E "gemm_ubench|pgemm" >> $O; done; done
for a in 0 32; do timeout 200 build_ub/gemm_ub_p$a 131072 384 1536 100 2>&1 | grep -E "gemm_ubench|pgemm" >> $O; done
cat $O
