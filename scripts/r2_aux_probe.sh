#!/bin/bash
# cache-policy bits of the scan's LDS-DMA loads (aux: 1 = sc0, 2 = nt, 16 = sc1): package power and rate
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/aux; rm -rf $OUT; mkdir -p $OUT
for f in $ROOT/build_ub/aux_*; do
  n=$(basename $f)
  python $ROOT/scripts/power_sampler.py $OUT/$n.log -- $f 10000000 384 2500 > $OUT/$n.txt 2>&1
  echo "$n: $(tail -1 $OUT/$n.txt | sed 's/.*avg/avg/') | $(grep '^R' $OUT/$n.log | sed -n 4p | sed 's/.*sclk clock level: 1: (\([0-9]*\)Mhz).*Power (W): \([0-9.]*\)/sclk \1 MHz power \2 W/')"
done
