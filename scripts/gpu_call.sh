#!/bin/bash
mkdir -p gpurun_out
S=$(date +%s)
timeout 900 python bench.py > gpurun_out/r5_bench_final.json 2> gpurun_out/r5_bench_final.err
echo "bench rc=$? wall $(( $(date +%s) - S )) s" > gpurun_out/r5_bench_final_summary.txt
python - >> gpurun_out/r5_bench_final_summary.txt <<'PY'
import json
d = json.load(open("gpurun_out/r5_bench_final.json"))
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "outside", d["ms_outside_collect_launch"])
print("host_api", d["host_api"]["value"])
print("ingest", d["ingest"]["value"], d["ingest"]["roofline"]["frac"], "bge", d["ingest_bge_base"]["value"], d["ingest_bge_base"]["roofline"]["frac"])
print("bf16x3", {k: (round(v["value"]), round(v["mfma_frac"], 3)) for k, v in d["ingest_bf16x3"].items()})
print("text_ingest", d["text_ingest"]["value"], "f32_rows", d["f32_rows"]["roofline"]["frac"])
print("qlat", {k: (round(v["encode_ms_p50"], 3)) for k, v in d["query_latency"].items() if isinstance(v, dict)})
PY
cat gpurun_out/r5_bench_final_summary.txt
