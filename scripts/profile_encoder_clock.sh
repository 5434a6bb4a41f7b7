#!/bin/bash
# effective shader clock and MFMA busy fraction per encoder kernel (PMC pass, --kernel-trace only)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for MODEL in l6 bge; do
echo "== $MODEL"
rm -rf /tmp/encpmc
timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/encpmc -- python $ROOT/scripts/gpu_encoder_prof.py $MODEL > /dev/null 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/encpmc/**/*_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0][-40:]
    acc[name][r["Counter_Name"]].append((float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
for name, c in acc.items():
    if "GRBM_GUI_ACTIVE" not in c: continue
    g = c["GRBM_GUI_ACTIVE"]; m = c.get("SQ_VALU_MFMA_BUSY_CYCLES", [])
    n = len(g); gui = sum(v for v, _ in g) / n / 8.0; dur = sum(d for _, d in g) / n
    mf = sum(v for v, _ in m) / max(1, len(m))
    print(f"{name:42s} n={n:4d} dur={dur:8.1f} us  clk={gui/dur/1e3:5.2f} GHz  mfma_busy={mf/(1024*gui)*100 if gui else 0:5.1f}%")
PY
done
