#!/bin/bash
# round 6: SQ counters of attention_x3_kernel (MX_PREC_MIXED, bge-base and MiniLM-L6 shapes): where a wave's cycles go.
# PMC passes with --kernel-trace only; per-launch means.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for M in bge l6; do
for C in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS"; do
rm -rf /tmp/x3pmc
timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/x3pmc -- python $ROOT/scripts/gpu_encoder_prof.py $M mixed > /dev/null 2>&1
python3 - "$M" <<PY
import csv, glob, collections, sys
fs = glob.glob("/tmp/x3pmc/**/*_counter_collection.csv", recursive=True)
if not fs:
    print("no counter file for: $C"); raise SystemExit
acc = collections.defaultdict(list); dur = []
for r in csv.DictReader(open(fs[0])):
    if "attention_x3" not in r["Kernel_Name"]: continue
    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(sys.argv[1], " ".join(f"{k}={sum(v)/len(v):.4g}" for k, v in sorted(acc.items())), f"dur_us={sum(dur)/max(1,len(dur)):.1f} n={len(dur)}")
PY
done
done
