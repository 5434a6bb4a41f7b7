#!/bin/bash
# SQ counters of the encoder kernels (attention in particular): where do the wave-cycles go?
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/attn_pmc_${MODEL:-bge}${TAG}; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -- python $ROOT/scripts/gpu_encoder_prof.py ${MODEL:-bge} > /dev/null 2> $OUT/p$i.log
done
python - $OUT <<'PY'
import csv, glob, os, sys
acc = {}
for path in glob.glob(os.path.join(sys.argv[1], "p*", "**", "*_counter_collection.csv"), recursive=True):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"]
            name = "attention" if "attention" in k else "tail" if "tail_kernel" in k else ("pgemm" + k[k.find("<") : k.find("<") + 3]) if "pgemm_kernel" in k else "ln_rows" if "ln_rows" in k else None
            if not name: continue
            a = acc.setdefault((name, row["Counter_Name"]), {})
            a[row["Dispatch_Id"]] = a.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
names = sorted({n for n, _ in acc})
for n in names:
    c = {cn: sum(v.values()) / len(v) for (nn, cn), v in acc.items() if nn == n}
    wc = c.get("SQ_WAVE_CYCLES", 1)
    print(n, {k: round(v / wc, 3) for k, v in c.items() if k.startswith("SQ_WAIT") or k.startswith("SQ_ACTIVE")},
          "mfma_busy/(1024*cycles)=%.3f" % (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024.0 * c.get("GRBM_GUI_ACTIVE", 8) / 8.0)),
          "lds_conflict/idx=%.3f" % (c.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, c.get("SQ_LDS_IDX_ACTIVE", 1))),
          "valu_insts=%.3g lds_insts=%.3g" % (c.get("SQ_INSTS_VALU", 0), c.get("SQ_INSTS_LDS", 0)))
PY
