#!/bin/bash
# Per-step structure of the search bench from a rocprofv3 kernel trace: kernel durations and the gaps
# between them (median over the timed steps of the headline leg).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/step_gaps; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python $ROOT/bench.py --ingest-chunks 0 --no-cpu-baseline --side-steps 0 --alt-steps 0 --steps 40 --warmup 10 > $OUT/bench.json 2> $OUT/log.txt
python - $OUT <<'PY'
import csv, glob, statistics, sys, os
f = glob.glob(os.path.join(sys.argv[1], "t", "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
K = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
idx = [i for i, (n, _, _) in enumerate(K) if "scan16_kernel<3, 1>" in n]
def nm(n):
    for t in ("prep_queries", "scan16_kernel<3, 0>", "theta", "scan16_kernel<3, 1>", "finish", "copyBuffer", "fillBuffer"):
        if t in n: return t
    return n[:30]
steps = []
for a, b in zip(idx[12:48], idx[13:49]):
    seg = K[a:b + 1]
    d = {"collect": (seg[0][2] - seg[0][1]) / 1e3, "period": (seg[-1][1] - seg[0][1]) / 1e3}
    for j in range(1, len(seg)):
        n = nm(seg[j][0])
        if j < len(seg) - 1: d[n] = (seg[j][2] - seg[j][1]) / 1e3
        d["gap_before_" + n] = (seg[j][1] - seg[j - 1][2]) / 1e3
    steps.append(d)
for k in steps[0]:
    v = [s[k] for s in steps if k in s]
    print(f"{k:40s} median {statistics.median(v):9.1f} us")
PY
