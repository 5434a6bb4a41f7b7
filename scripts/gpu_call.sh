#!/bin/bash
mkdir -p gpurun_out
S=$(date +%s)
timeout 900 python bench.py > gpurun_out/r5_bench_default_v2.json 2> gpurun_out/r5_bench_default_v2.err
echo "bench wall $(( $(date +%s) - S )) s rc=$?" > gpurun_out/r5_full_check.txt
tail -2 gpurun_out/r5_bench_default_v2.err >> gpurun_out/r5_full_check.txt
S=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 >> gpurun_out/r5_full_check.txt
echo "pytest wall $(( $(date +%s) - S )) s" >> gpurun_out/r5_full_check.txt
python - >> gpurun_out/r5_full_check.txt <<'PY'
import json
d = json.load(open("gpurun_out/r5_bench_default_v2.json"))
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "outside", d["ms_outside_collect_launch"])
print("enc_like", {k: d["enc_like_10M"].get(k) for k in ("value", "retry_queries", "fallback_queries", "ids_equal_exact_path", "filter_centred")})
print("text_ingest", {k: d["text_ingest"].get(k) for k in ("value", "windows_per_s", "text_MBps", "errors", "query_finds_its_window", "error")}, d["text_ingest"].get("segmenter"))
print("ingest", d["ingest"]["value"], d["ingest"]["roofline"]["frac"], "bge", d["ingest_bge_base"]["value"], d["ingest_bge_base"]["roofline"]["frac"])
print("f32_rows", d["f32_rows"]["value"], d["f32_rows"]["roofline"]["frac"])
print("qlat", d["query_latency"]["all-MiniLM-L6-v2"], d["query_latency"]["all-MiniLM-L12-v2"])
PY
cat gpurun_out/r5_full_check.txt
