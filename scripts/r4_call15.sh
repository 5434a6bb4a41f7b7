#!/bin/bash
mkdir -p gpurun_out; out=$GRAFT_REPO_ROOT/gpurun_out/r4_attn6.txt; : > $out
cd /tmp && export TMPDIR=/tmp
run() {  # model skew
rm -rf /tmp/st; MEMEX_HIP_ATTN_SKEW=$2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python $GRAFT_REPO_ROOT/scripts/gpu_encoder_prof.py $1 > /dev/null 2>&1
python - $1 $2 >> $out <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/st/**/*_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "attention" in r["Name"]: print("%s SKEW=%s" % tuple(sys.argv[1:3]), r["Name"].split("(")[0][:40], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
}
for m in bge l6; do for sk in 0 1 2 3; do run $m $sk; done; done
cat $out
