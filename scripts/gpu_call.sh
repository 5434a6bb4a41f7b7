#!/bin/bash
mkdir -p gpurun_out
S=$(date +%s)
timeout 900 python bench.py > gpurun_out/r5_bench_final.json 2> gpurun_out/r5_bench_final.err
echo "bench rc=$? wall $(( $(date +%s) - S )) s" > gpurun_out/r5_final_check.txt
timeout 2400 python -m pytest tests -m gpu -x -q > /tmp/pt.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5_final_check.txt
grep -E "passed|failed|Error" /tmp/pt.log | tail -4 >> gpurun_out/r5_final_check.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> gpurun_out/r5_final_check.txt
python - >> gpurun_out/r5_final_check.txt <<'PY'
import json
d = json.load(open("gpurun_out/r5_bench_final.json"))
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "outside", d["ms_outside_collect_launch"])
print("host_api", d["host_api"]["value"], d["host_api"]["ms_per_step"])
print("small", {k: round(v["ms_per_call"], 3) for k, v in d["small_batches"].items()})
print("enc_like", {k: d["enc_like_10M"].get(k) for k in ("value", "retry_queries", "fallback_queries", "ids_equal_exact_path")})
print("text_ingest", {k: d["text_ingest"].get(k) for k in ("value", "windows_per_s", "errors", "query_finds_its_window", "error")})
print("ingest", d["ingest"]["value"], d["ingest"]["roofline"]["frac"], "bge", d["ingest_bge_base"]["value"], d["ingest_bge_base"]["roofline"]["frac"])
print("bf16x3", {k: (round(v["value"]), round(v["mfma_frac"], 3)) for k, v in d["ingest_bf16x3"].items()})
print("f32_rows", d["f32_rows"]["value"], d["f32_rows"]["roofline"]["frac"])
print("qlat", d["query_latency"]["all-MiniLM-L6-v2"], d["query_latency"]["all-MiniLM-L12-v2"])
PY
cat gpurun_out/r5_final_check.txt
