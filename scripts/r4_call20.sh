#!/bin/bash
# attention: heads (head pairs at d = 32) walked per workgroup, uniform 512-token sequences
mkdir -p gpurun_out; out=$GRAFT_REPO_ROOT/gpurun_out/r4_attn_hpw.txt; : > $out
cd /tmp && export TMPDIR=/tmp
run() {  # model hpw
rm -rf /tmp/st; MEMEX_HIP_ATTN_HPW=$2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python $GRAFT_REPO_ROOT/scripts/gpu_encoder_prof.py $1 > /dev/null 2>&1
python - $1 $2 >> $out <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/st/**/*_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "attention" in r["Name"]: print("%s HPW=%s" % tuple(sys.argv[1:3]), r["Name"].split("(")[0][:44], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
}
for h in 1 2 3 4 6 12; do run bge $h; done
for h in 1 2 3 6; do run l6 $h; done
cat $out
