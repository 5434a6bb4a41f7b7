#!/bin/bash
# round 3, GPU call 6: int8 filter copy (scan8.hip): parity, then throughput against the bf16 copy
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/c6; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
( time timeout 900 python -m pytest tests/test_search_gpu.py -m gpu -q -x ) > "$OUT/pytest.log" 2>&1
grep -E "passed|failed" "$OUT/pytest.log" | tail -3; grep -E "^E  " "$OUT/pytest.log" | head -20
for F in bf16 i8; do
  echo "== MEMEX_HIP_FILTER=$F"
  timeout 600 python bench.py --scan $F --steps 20 --warmup 5 --ingest-chunks 0 --no-cpu-baseline --alt-steps 0 --side-steps 0 > "$OUT/bench_$F.json" 2> "$OUT/bench_$F.err"
  python - "$OUT/bench_$F.json" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        d = json.loads(ln)
        print("value", round(d["value"]), "ms/step", round(d["ms_per_step"], 4), "roofline", {k: d["roofline"].get(k) for k in ("kernel", "achieved", "frac", "ms_per_launch", "bytes_per_launch")},
              "cand/q", d["candidates_per_query"], "e1", d.get("approx_err_bound"), "fallback", d["fallback_queries"], "retry", d["retry_queries"], "outside", round(d["ms_outside_collect_launch"], 4))
PY
  tail -3 "$OUT/bench_$F.err"
done
