#!/bin/bash
# quick GPU check: search parity tests, then per-kernel averages of a short headline run (rocprofv3 --stats)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-q}
OUT=$ROOT/gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd $ROOT
python -m pytest tests/test_search_gpu.py tests/test_sharded_gpu.py -x -q 2>&1 | tail -15 > "$OUT/pytest.log"
cd /tmp && export TMPDIR=/tmp
for DIV in ${DIVS:-64}; do
  export MEMEX_HIP_DEBUG=sample_div=$DIV
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$DIV" -- python $ROOT/bench.py --ingest-chunks 0 --no-cpu-baseline --steps 40 --warmup 5 --alt-steps 0 --side-steps ${SIDE:-0} > "$OUT/bench_$DIV.json" 2> "$OUT/bench_$DIV.err"
  python - "$OUT/stats_$DIV" <<'PY'
import csv, glob, sys, os
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True)
if f:
    for r in list(csv.DictReader(open(f[0])))[:12]:
        if "at::native" in r["Name"]: continue
        print(f"{r['Name'][:64]:64s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
  python - "$OUT/bench_$DIV.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print({k: d[k] for k in ("value", "ms_per_step", "ms_outside_collect_launch", "candidates_per_query", "retry_queries", "fallback_queries", "approx_err_bound")}, d["roofline"]["frac"], d["roofline"]["ms_per_launch"])
    for k in ("clustered", "host_api", "cfg4_shard_10Mx768"):
        if k in d: print(k, {x: d[k][x] for x in ("value", "ms_per_step", "candidates_per_query", "retry_queries", "fallback_queries")}, d[k]["roofline"]["frac"], d[k]["roofline"]["ms_per_launch"])
except Exception as e:
    print("bench parse failed", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-2000:])
PY
done
cat "$OUT/pytest.log" | tail -5
