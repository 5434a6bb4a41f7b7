#!/bin/bash
cd /tmp && export TMPDIR=/tmp
ROOT="$GRAFT_REPO_ROOT"; OUT=$ROOT/gpurun_out/r5ac; mkdir -p $OUT
for cfgs in "8192 128" "8192 128 ragged" "16384 64" "4096 256"; do
  tag=$(echo $cfgs | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$tag" -- python $ROOT/scripts/gpu_enc_short_prof.py $cfgs > /dev/null 2> "$OUT/$tag.log"
  f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1); echo "== all-MiniLM-L12-v2 shape, B S = $cfgs"
  python - "$f" <<'P'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:7]:
    print(f"  {r['Name'].split('(')[0][-44:]:44s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['Percentage']):5.1f} %")
P
done > $ROOT/gpurun_out/r5ac_short_window_kernels_after.txt
cat $ROOT/gpurun_out/r5ac_short_window_kernels_after.txt; rm -rf $OUT
