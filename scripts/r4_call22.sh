#!/bin/bash
# persistent attention: d = 32 two heads per stage against one, workgroups per grid
mkdir -p gpurun_out; out=$GRAFT_REPO_ROOT/gpurun_out/r4_attn_persist2.txt; : > $out
cd /tmp && export TMPDIR=/tmp
run() {  # model pair cus
rm -rf /tmp/st; MEMEX_HIP_ATTN_PAIR=$2 MEMEX_HIP_ATTN_CUS=$3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python $GRAFT_REPO_ROOT/scripts/gpu_encoder_prof.py $1 > /dev/null 2>&1
python - $1 $2 $3 >> $out <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/st/**/*_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "attention" in r["Name"]: print("%s PAIR=%s CUS=%s" % tuple(sys.argv[1:4]), r["Name"].split("(")[0][:44], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
}
run l6 1 0; run l6 0 0; run l6 1 0; run l6 0 0
run bge 1 0; run bge 1 512; run l6 1 512
cat $out
