#!/bin/bash
# r6a: this round's baseline of the GEMM / tail microbenchmarks (512-register tail2_kernel next to tail_kernel)
cd /tmp && export TMPDIR=/tmp
ROOT="$GRAFT_REPO_ROOT"; O=$ROOT/gpurun_out
{
  echo "== tail_ub (tail_kernel: 2 waves/SIMD, 64 rows per workgroup | tail2_kernel: 1 wave/SIMD, 512 registers, 128 rows per workgroup)"
  timeout 300 $ROOT/build_ub/tail_ub 131072 1536 3000 1 | grep -E "^tail|checksum|max err"
  echo "== gemm_ub hidden 768"
  timeout 300 $ROOT/build_ub/gemm_ub 131072 768 3072 200
  echo "== gemm_ub hidden 384"
  timeout 300 $ROOT/build_ub/gemm_ub 131072 384 1536 400
} > $O/r6a_ubench_baseline.txt 2>&1
cat $O/r6a_ubench_baseline.txt
