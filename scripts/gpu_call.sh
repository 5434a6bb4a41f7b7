#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r5_precise_final.txt
: > $O
timeout 1500 python -m pytest tests/test_encoder_gpu.py tests/test_pretrained.py tests/test_pipeline_native_gpu.py -m gpu -x -q > /tmp/pt.log 2>&1
echo "pytest rc=$?" >> $O
grep -E "passed|failed|Error" /tmp/pt.log | tail -4 >> $O
timeout 600 python scripts/gpu_encoder_precise.py 2>&1 | grep chunks >> $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in l6 bge; do
  rm -rf /tmp/prof_p
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_p -- python $R/scripts/gpu_encoder_prof.py $m bf16x3 > /tmp/prof.log 2>&1
  f=$(find /tmp/prof_p -name "*kernel_stats.csv" | head -1)
  echo "== $m bf16x3" >> $R/$O
  python - "$f" >> $R/$O <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:7]:
    print(f"  {r['Name'][:60]:60s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['TotalDurationNs'])/tot*100:5.1f} %")
PY
done
cat $R/$O
