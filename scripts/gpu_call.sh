#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r5_gemm_shape_probe.txt
: > $O
for rep in 1 2; do for a in 0 4; do
  timeout 200 build_ub/gemm_ub_a$a 131072 384 1536 100 >> $O 2>&1
done; done
for a in 0 4; do timeout 200 build_ub/gemm_ub_a$a 131072 768 3072 50 >> $O 2>&1; done
cat $O
