#!/bin/bash
cd /tmp && export TMPDIR=/tmp
ROOT="$GRAFT_REPO_ROOT"; OUT=$ROOT/gpurun_out/r5aa; mkdir -p $OUT
for cfgs in "4096 128 full roberta" "4096 128 ragged roberta"; do
  tag=$(echo $cfgs | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$tag" -- python $ROOT/scripts/gpu_enc_short_prof.py $cfgs > /dev/null 2> "$OUT/$tag.log"
  f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1); echo "== B S = $cfgs"
  python - "$f" <<'P'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print(f"  {r['Name'].split('(')[0][-44:]:44s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['Percentage']):5.1f} %")
P
done > $ROOT/gpurun_out/r5aa_roberta_short.txt
cat $ROOT/gpurun_out/r5aa_roberta_short.txt; tail -3 $OUT/*.log | tail -5; rm -rf $OUT
