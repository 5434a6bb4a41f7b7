#!/bin/bash
# ln_rows_kernel: rows in forward / reverse order (does the Infinity Cache still hold what the GEMM wrote last?)
mkdir -p gpurun_out; out=$GRAFT_REPO_ROOT/gpurun_out/r4_ln_rev.txt; : > $out
cd /tmp && export TMPDIR=/tmp
for rev in 0 1 0 1; do
rm -rf /tmp/st; MEMEX_HIP_LN_REV=$rev timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python $GRAFT_REPO_ROOT/scripts/gpu_encoder_prof.py bge > /dev/null 2>&1
python - $rev >> $out <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/st/**/*_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "ln_rows" in r["Name"] or "pgemm_kernel<1>" in r["Name"] or "pgemm_kernel<2>" in r["Name"]: print("REV=%s" % sys.argv[1], r["Name"].split("(")[0][:44], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
done
cat $out
