#!/bin/bash
# finish_kernel stage timing: MEMEX_HIP_FINISH_STOP=1..4 returns after that stage (results garbage)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/${1:-fp}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for STOP in ${STOPS:-0 1 2 3 4}; do
  export MEMEX_HIP_FINISH_STOP=$STOP
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/s$STOP" -- python $ROOT/bench.py --ingest-chunks 0 --no-cpu-baseline --steps 30 --warmup 3 --alt-steps 0 --side-steps 0 --recall-queries 0 --rows ${ROWS:-10000000} > "$OUT/b$STOP.json" 2> "$OUT/b$STOP.err"
  python - "$OUT/s$STOP" $STOP <<'PY'
import csv, glob, sys, os
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True)
for r in csv.DictReader(open(f[0])):
    if any(t in r["Name"] for t in ("finish_kernel", "theta_kernel", "scan16_kernel", "scan8_kernel", "prep_queries")):
        print(f"stop={sys.argv[2]} {r['Name'][:50]:50s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
done
