#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_doc_pass.txt
: > $O
timeout 300 python $R/scripts/gpu_doc_pass_prof.py 72 128 >> $O 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_doc -- python $R/scripts/gpu_doc_pass_prof.py 72 128 > /tmp/prof.log 2>&1
f=$(find /tmp/prof_doc -name "*kernel_stats.csv" | head -1); ls -R /tmp/prof_doc | head -20 >> /tmp/prof.log
python - "$f" >> $O <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over 55 calls")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print(f"  {r['Name'][:60]:60s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['TotalDurationNs'])/tot*100:5.1f} %")
PY
cat $O
