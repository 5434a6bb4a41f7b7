#!/bin/bash
# PMC counters of the two fused-MLP kernels on the microbenchmark (build_ub/tail_ub): where do the
# wave-cycles go, and what does the weight stream cost in L1/L2?
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/tail_pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA" \
         "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
         "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCC_TAG_STALL_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 90 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -- $ROOT/build_ub/tail_ub 131072 1536 3 > /dev/null 2> $OUT/p$i.log
done
python - $OUT <<'PY'
import csv, glob, os, sys
acc = {}
for path in glob.glob(os.path.join(sys.argv[1], "p*", "**", "*_counter_collection.csv"), recursive=True):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"]
            name = "tail" if "tail_kernel" in k else None
            if not name: continue
            a = acc.setdefault((name, row["Counter_Name"]), {})
            a[row["Dispatch_Id"]] = a.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
for n in sorted({n for n, _ in acc}):
    c = {cn: sum(v.values()) / len(v) for (nn, cn), v in acc.items() if nn == n}
    wc = c.get("SQ_WAVE_CYCLES", 1)
    print(n, {k: round(v / wc, 3) for k, v in c.items() if k.startswith("SQ_WAIT") or k.startswith("SQ_ACTIVE")})
    print("   mfma_busy/(1024*cycles)=%.3f" % (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024.0 * c.get("GRBM_GUI_ACTIVE", 8) / 8.0)),
          "lds_conflict/idx=%.3f" % (c.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, c.get("SQ_LDS_IDX_ACTIVE", 1))))
    print("   raw:", {k: "%.4g" % v for k, v in sorted(c.items())})
PY
