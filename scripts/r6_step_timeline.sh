#!/bin/bash
# round 6: the headline step as a timeline -- the five launches of a 256-query batch on the int8 copy, their durations and the gaps
# between them (rocprofv3 --kernel-trace; medians over 36 timed steps), plus the same for the enc_like leg (centred copy).
# What VERDICT r5 #5 asked for in lieu of <= 0.17 ms outside the collect launch: where those microseconds are.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/step_timeline; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python $ROOT/bench.py --ingest-chunks 0 --bge-chunks 0 --short-seqs 0 --no-cpu-baseline --side-steps 0 --alt-steps 0 --small-steps 0 --shard-legs 0 --enc-like-rows 0 --cfg2-segments 0 --text-docs 0 --precise-chunks 0 --steps 60 --warmup 10 > $OUT/bench.json 2> $OUT/log.txt
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/e -- python $ROOT/scripts/gpu_enc_like.py 10000000 60 > $OUT/enc_like.json 2> $OUT/log_e.txt
python3 - $OUT <<'PY'
import csv, glob, statistics, sys, os
def timeline(d, collect):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    K = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
    idx = [i for i, (n, _, _) in enumerate(K) if collect in n]
    def nm(n):
        for t in ("prep_queries", "scan8_kernel<3, 0", "theta", "scan8_kernel<3, 1", "finish", "copyBuffer", "fillBuffer"):
            if t in n: return {"scan8_kernel<3, 0": "sample scan", "scan8_kernel<3, 1": "collect scan"}.get(t, t)
        return n[:30]
    steps = []
    for a, b in zip(idx[14:50], idx[15:51]):
        seg = K[a:b + 1]
        d_ = {"collect scan": (seg[0][2] - seg[0][1]) / 1e3, "period (collect start to collect start)": (seg[-1][1] - seg[0][1]) / 1e3}
        for j in range(1, len(seg)):
            n = nm(seg[j][0])
            if j < len(seg) - 1: d_[n] = (seg[j][2] - seg[j][1]) / 1e3
            d_["  gap before " + n] = (seg[j][1] - seg[j - 1][2]) / 1e3
        steps.append(d_)
    for k in steps[0]:
        v = [s[k] for s in steps if k in s]
        print(f"  {k:44s} median {statistics.median(v):9.1f} us")
    per = statistics.median([s["period (collect start to collect start)"] for s in steps]); col = statistics.median([s["collect scan"] for s in steps])
    print(f"  outside the collect launch: {per - col:.1f} us of a {per:.1f} us period")
print("== headline: 10M x 384 Gaussian rows, plain int8 copy"); timeline(os.path.join(sys.argv[1], "t"), "scan8_kernel<3, 1, 1, false>")
print("== enc_like_10M: centred int8 copy"); timeline(os.path.join(sys.argv[1], "e"), "scan8_kernel<3, 1, 1, true>")
PY
rm -rf $OUT/t $OUT/e
