#!/bin/bash
# round-2 probe 1: power/clock evidence for the scan16 kernel and its ablations, SQ stall counters.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r2p1
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters_list.txt" 2>&1
rocm-smi --showpower --showclocks --showmaxpower > "$OUT/smi_idle.txt" 2>&1
ls /sys/class/drm/card*/device/hwmon/hwmon*/ > "$OUT/hwmon_ls.txt" 2>&1
for V in 0 1 2 4; do
  python $ROOT/scripts/power_sampler.py "$OUT/power_scan16_ub_$V.log" -- $ROOT/build_ub/scan16_ub_$V 10000000 384 3000 > "$OUT/ub_$V.txt" 2>&1
done
python $ROOT/scripts/power_sampler.py "$OUT/power_scan16_ub_0_768.log" -- $ROOT/build_ub/scan16_ub_0 10000000 768 1500 > "$OUT/ub_0_768.txt" 2>&1
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  for V in 0 4; do
    D=$OUT/pmc${i}_ub$V
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$D" -- $ROOT/build_ub/scan16_ub_$V 10000000 384 8 > "$D.txt" 2>&1
  done
done
# reduce: per counter mean over dispatches of the scan kernel
python - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
res = {}
for d in sorted(glob.glob(os.path.join(out, "pmc*_ub*"))):
    if not os.path.isdir(d): continue
    for path in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        acc = {}
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if "scan16_kernel" not in row["Kernel_Name"]: continue
                a = acc.setdefault(row["Counter_Name"], {})
                a[row["Dispatch_Id"]] = a.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
        for c, per in acc.items():
            v = sorted(per.values())
            res[(os.path.basename(d), c)] = (sum(v) / len(v), len(v))
with open(os.path.join(out, "pmc_summary.txt"), "w") as g:
    for (d, c), (m, n) in sorted(res.items()):
        g.write(f"{d:14s} {c:34s} mean/launch {m:16.1f} over {n} launches\n")
print(open(os.path.join(out, "pmc_summary.txt")).read())
PY
tail -3 "$OUT"/ub_*.txt
