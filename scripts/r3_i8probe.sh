#!/bin/bash
# round 3: would an 8-bit filter copy pay?  scan16_kernel on the same 8 KiB slots with the bf16 MFMA (0) and with
# v_mfma_i32_32x32x32_i8 (1; values meaningless): launch time and package power at 384 and 768 dims
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/i8probe
for v in 0 1; do
  for ds in 384 768; do
    python scripts/power_sampler.py gpurun_out/i8probe/power_${v}_$ds.log -- build_ub/scan16_i8probe_$v 10000000 $ds 2000 2>&1 | grep "^scan16"
    grep "^S" gpurun_out/i8probe/power_${v}_$ds.log | sed -E 's/.*power1_input=([0-9]+) freq1_input=([0-9]+).*/\1 \2/' | awk '{n++; if (n>40) {p+=$1/1e6; c+=$2/1e6; m++}} END {if (m) printf "   power %.0f W  clock %.0f MHz (%d hwmon samples after 2 s)\n", p/m, c/m, m}'
  done
done
