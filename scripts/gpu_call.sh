#!/bin/bash
mkdir -p gpurun_out
T0=$(date +%s); python bench.py > gpurun_out/r5_bench_final.json 2> gpurun_out/r5_bench_final.err; echo "bench wall seconds: $(( $(date +%s) - T0 ))"
tail -3 gpurun_out/r5_bench_final.err
python -c "
import json; d=json.load(open('gpurun_out/r5_bench_final.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['f32_rows']['roofline']['frac'], d['enc_like_10M']['value'], d['cfg2']['retry_queries'], d['ingest']['roofline']['frac'], d['ingest_bge_base']['roofline']['frac'], d['query_latency']['all-MiniLM-L6-v2']['encode_ms_p50'], d['query_latency']['all-MiniLM-L12-v2']['encode_ms_p50'], d['shard_1p25Mx384'].get('predicted_n8_qps'))"
python -c "import __graft_entry__ as g; g.smoke()"
