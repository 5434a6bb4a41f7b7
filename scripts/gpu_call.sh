#!/bin/bash
# r6k: scan8_kernel alone: production against "half the fragment reads" (what a half-tile split of the waves would buy)
cd /tmp && export TMPDIR=/tmp
ROOT="$GRAFT_REPO_ROOT"
{
for rep in 1 2; do
for ab in 0 2; do
  timeout 120 $ROOT/build_ub/scan8_ub_$ab 10000000 384 200 256
  timeout 120 $ROOT/build_ub/scan8_ub_$ab 10000000 768 100 256
done; done
} > $ROOT/gpurun_out/r6k_scan8_half_reads.txt 2>&1
cat $ROOT/gpurun_out/r6k_scan8_half_reads.txt
