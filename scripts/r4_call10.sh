#!/bin/bash
# attention kernels: how much of a launch is staging + stores (MEMEX_HIP_ATTN_SAFE=2 skips the key loop)?
mkdir -p gpurun_out; out=gpurun_out/r4_attn_staging.txt; : > $out
cd /tmp && export TMPDIR=/tmp
for mode in 0 2; do for model in bge l6; do
rm -rf /tmp/st; MEMEX_HIP_ATTN_SAFE=$mode rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python $GRAFT_REPO_ROOT/scripts/gpu_encoder_prof.py $model > /dev/null 2>&1
python - $mode $model >> $GRAFT_REPO_ROOT/$out <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/st/**/*_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "attention" in r["Name"]: print("MEMEX_HIP_ATTN_SAFE=%s %s" % (sys.argv[1], sys.argv[2]), r["Name"].split("(")[0][:40], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
done; done
cat $GRAFT_REPO_ROOT/$out
