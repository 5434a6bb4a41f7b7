#!/bin/bash
# the --kernel-trace --stats pass of scripts/profile_search.sh alone (10M x 384 with its side legs, without the
# small-batch legs): refreshes profiles/r3_bench_kernel_stats.csv and r3_bench_under_rocprof.json
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/stats_r3; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python $ROOT/bench.py --ingest-chunks 0 --no-cpu-baseline --steps 50 --warmup 10 --alt-steps 20 --side-steps 20 --small-steps 0 > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log"
F=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1)
cp "$F" "$OUT/r3_bench_kernel_stats.csv"
grep "^{" "$OUT/bench_under_rocprof.json" | tail -1 > "$OUT/r3_bench_under_rocprof.json"
python - "$OUT" <<'PY'
import csv, json, sys, os
o = sys.argv[1]
d = json.load(open(os.path.join(o, "r3_bench_under_rocprof.json")))
print("bench under rocprof: QPS", round(d["value"]), "collect ms by HIP events", round(d["roofline"]["ms_per_launch"], 4))
for r in csv.DictReader(open(os.path.join(o, "r3_bench_kernel_stats.csv"))):
    if any(k in r["Name"] for k in ("scan8_kernel<3", "scan16_kernel<3", "finish", "scan8_kernel<6")):
        print(f"{r['Name'][:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
rm -rf "$OUT/stats"
