#!/bin/bash
# round 3: int8 filter copy, sample size sweep (MEMEX_HIP_DEBUG sample_div=N): step time, collect launch, candidates, retries
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/i8s
for DIV in ${DIVS:-64 32 16 8 4}; do
  MEMEX_HIP_DEBUG=sample_div=$DIV timeout 600 python bench.py --scan ${SCAN:-i8} --steps 20 --warmup 5 --ingest-chunks 0 --no-cpu-baseline --alt-steps 0 --side-steps 0 > gpurun_out/i8s/b_$DIV.json 2> gpurun_out/i8s/b_$DIV.err
  python - gpurun_out/i8s/b_$DIV.json $DIV <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        d = json.loads(ln)
        print("div", sys.argv[2], "QPS", round(d["value"]), "ms/step", round(d["ms_per_step"], 4), "collect ms", round(d["roofline"]["ms_per_launch"], 4), "frac", round(d["roofline"]["frac"], 3),
              "cand/q", round(d["candidates_per_query"], 1), "e1", round(d.get("approx_err_bound") or 0, 5), "retry", d["retry_queries"], "outside", round(d["ms_outside_collect_launch"], 4))
PY
done
