#!/bin/bash
# pgemm_kernel: m-tiles walked from the end for some launch kinds (does a consumer find its producer's last output in the Infinity Cache?)
mkdir -p gpurun_out; out=$GRAFT_REPO_ROOT/gpurun_out/r4_pgemm_rev.txt; : > $out
cd /tmp && export TMPDIR=/tmp
for rev in 0 16 8 24 4 1 3 0; do
rm -rf /tmp/st; MEMEX_HIP_PGEMM_REV=$rev timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python $GRAFT_REPO_ROOT/scripts/gpu_encoder_prof.py bge > /dev/null 2>&1
python - $rev >> $out <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/st/**/*_kernel_stats.csv", recursive=True)[0]
tot = 0.0; parts = []
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0].replace("void mx::", "")
    if any(k in n for k in ("pgemm", "ln_rows", "attention")):
        parts.append("%s %.1f" % (n[:22], float(r["AverageNs"]) / 1e3)); tot += float(r["TotalDurationNs"]) / 1e3 / 144
print("REV=%-3s layer %.1f us |" % (sys.argv[1], tot), " | ".join(sorted(parts)))
PY
done
cat $out
