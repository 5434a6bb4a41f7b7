#!/bin/bash
# round 3, GPU call 7: whole GPU suite with the int8 filter copy as the default, then the default bench line
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/c7; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
( time timeout 2400 python -m pytest tests -m gpu -q ) > "$OUT/pytest.log" 2>&1
grep -E "passed|failed" "$OUT/pytest.log" | tail -3; grep -E "^(FAILED|ERROR)|^E  " "$OUT/pytest.log" | head -30
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 400 "$OUT/bench.err"
python - "$OUT/bench.json" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        d = json.loads(ln)
        print("value", round(d["value"]), "ms/step", round(d["ms_per_step"], 4), d["roofline"]["kernel"], "frac", round(d["roofline"]["frac"], 3), "ms", round(d["roofline"]["ms_per_launch"], 4),
              "cand/q", round(d["candidates_per_query"], 1), "retry", d["retry_queries"], "fallback", d["fallback_queries"], "outside", round(d["ms_outside_collect_launch"], 4), "recall", d["recall_at_10"], d["ids_equal_exact_path"])
        for k in ("other_scan", "host_api", "clustered", "anisotropic", "cfg4_shard_10Mx768", "cfg2"):
            v = d.get(k)
            if v: print(k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk in ("scan", "value", "ms_per_step", "ids_equal_main_run", "candidates_per_query", "retry_queries", "fallback_queries", "search_value", "embed_segments_per_s")})
PY
