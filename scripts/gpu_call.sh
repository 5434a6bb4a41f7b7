#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r5_precise_bigtile.txt
: > $O
for m in unset 8 9; do
  if [ $m = unset ]; then unset MEMEX_HIP_GEMM_BIG; else export MEMEX_HIP_GEMM_BIG=$m; fi
  echo "GEMM_BIG=$m" >> $O
  timeout 300 python -c "
import json, bench
r = bench.precise_ingest_leg(16384, 0); print({k: round(v['value']) for k, v in r.items()})
" 2>&1 | tail -1 >> $O
done
cat $O
