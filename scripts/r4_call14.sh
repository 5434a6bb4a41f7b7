#!/bin/bash
mkdir -p gpurun_out; out=$GRAFT_REPO_ROOT/gpurun_out/r4_attn5.txt; : > $out
cd /tmp && export TMPDIR=/tmp
run() {  # model safe hpw
rm -rf /tmp/st; MEMEX_HIP_ATTN_SAFE=$2 MEMEX_HIP_ATTN_HPW=$3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python $GRAFT_REPO_ROOT/scripts/gpu_encoder_prof.py $1 > /dev/null 2>&1
python - $1 $2 $3 >> $out <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/st/**/*_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "attention" in r["Name"]: print("%s SAFE=%s HPW=%s" % tuple(sys.argv[1:4]), r["Name"].split("(")[0][:40], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
}
for m in bge l6; do for safe in 0 2 3 4 5 6; do run $m $safe 0; done; done
cat $out
