#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_sharded_gpu.py tests/test_abi.py -m gpu -x -q -k "small_pass or regimes or sharded or abi or shard" 2>&1 | tail -8 > gpurun_out/r5_tests6.txt
cat gpurun_out/r5_tests6.txt
timeout 900 python bench.py --steps 20 --ingest-chunks 0 --bge-chunks 0 --no-cpu-baseline --enc-like-rows 0 --cfg2-segments 0 --alt-steps 0 --small-steps 0 > gpurun_out/r5_bench_shards.json 2> gpurun_out/r5_bench_shards.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5_bench_shards.json"))
for k in ("shard_1p25Mx384", "shard_1p25Mx768"):
    x = d[k]; print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in x.items() if a in ("ms_per_step", "eight_logical_shards_ms_per_step", "shard_local_ms", "serial_tail_ms", "exchange_and_merge_ms", "predicted_n8_qps", "predicted_n8_qps_no_exchange", "exchange_and_merge_error")})
PY
