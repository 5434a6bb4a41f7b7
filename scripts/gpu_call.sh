#!/bin/bash
# tail_kernel alone: back-to-back launches (warm inputs) against launches with 1 GB streamed in between (cold x / ctx), and
# the same gaps with x and ctx re-touched last (warm inputs, same clock history)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{ for a in "0 0" "1024 0" "1024 1" "1024 0" "1024 1"; do timeout 120 ./build_ub/tail_ub_a0 131072 1536 400 1 0 8 512 3 $a 2>&1 | grep -E "^tail "; done; } > gpurun_out/r5j_tail_cold_inputs.txt
cat gpurun_out/r5j_tail_cold_inputs.txt
