#!/bin/bash
# round 6: effective clock, MFMA-busy share and wait share of pgemm_kernel (8 waves, 2 per SIMD, 256 registers) and pgemm4_kernel
# (4 waves, 1 per SIMD, 512 registers) on one bge-base layer's shapes (scripts/gemm_ubench.hip), PMC pass with --kernel-trace only
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for BIN in gemm_ub_p4_0 gemm_ub_a3; do
echo "== $BIN (a3 = neither DMA in the loop nor epilogue)"
rm -rf /tmp/p4pmc
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/p4pmc -- $ROOT/build_ub/$BIN 131072 768 3072 30 > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
fs = glob.glob("/tmp/p4pmc/**/*_counter_collection.csv", recursive=True)
if not fs:
    print("no counter file"); raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    name = r["Kernel_Name"].split("(")[0]
    if "pgemm" not in name: continue
    name = name[-34:]
    acc[name][r["Counter_Name"]].append((float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
for name, c in sorted(acc.items()):
    if "GRBM_GUI_ACTIVE" not in c: continue
    g = c["GRBM_GUI_ACTIVE"]; n = len(g); gui = sum(v for v, _ in g) / n / 8.0; dur = sum(d for _, d in g) / n
    avg = lambda k: sum(v for v, _ in c.get(k, [(0, 0)])) / max(1, len(c.get(k, [1])))
    mf, wi, wc = avg("SQ_VALU_MFMA_BUSY_CYCLES"), avg("SQ_WAIT_INST_ANY"), avg("SQ_WAVE_CYCLES")
    print(f"{name:36s} n={n:4d} dur={dur:8.1f} us  clk={gui/dur/1e3:5.2f} GHz  mfma_busy={mf/(1024*gui)*100 if gui else 0:5.1f}%  wait_inst_any/wave_cycles={wi/wc*100 if wc else 0:5.1f}%")
PY
done
