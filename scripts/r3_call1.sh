#!/bin/bash
# round 3, GPU call 1: new tests (cfg2 end to end, sharded helpers / rollback, encoder after the workspace change),
# tail-kernel phase trace + start-up skew sweep, bench.py with the new legs, in-library multi-shard wiring check
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/c1; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
( time timeout 1500 python -m pytest tests/test_cfg2_gpu.py tests/test_sharded_gpu.py tests/test_encoder_gpu.py tests/test_persistence_gpu.py tests/test_cpp_host.py tests/test_pipeline_native_gpu.py -m gpu -x -q ) > "$OUT/pytest.log" 2>&1
tail -5 "$OUT/pytest.log"
{
  timeout 120 $ROOT/build_ub/tail_ub_tr 131072 1536 1500 1 0
  for it in 2 4 6 8 12; do timeout 120 $ROOT/build_ub/tail_ub 131072 1536 1500 1 $it 8 512 | grep "^tail"; done
  timeout 120 $ROOT/build_ub/tail_ub 131072 1536 1500 1 6 0 512 | grep "^tail"
  timeout 120 $ROOT/build_ub/tail_ub 131072 1536 1500 1 6 3 512 | grep "^tail"
  timeout 120 $ROOT/build_ub/tail_ub_tr 131072 1536 1500 1 6 8 512
} > "$OUT/tail.log" 2>&1
cat "$OUT/tail.log"
( time timeout 900 python bench.py --steps 20 --warmup 5 --hnsw-rows 0 --cpu-seconds 3 ) > "$OUT/bench.log" 2> "$OUT/bench.err"
tail -c 3000 "$OUT/bench.err"
python - "$OUT/bench.log" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        d = json.loads(ln)
        print("value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "outside", d["ms_outside_collect_launch"])
        for k in ("cfg2", "ingest_bge_base", "ingest"):
            print(k, json.dumps(d.get(k))[:900])
PY
( MEMEX_BENCH_ONE_DEVICE=1 MEMEX_HIP_SHARD_THREADS=1 timeout 600 python bench.py --gpus 4 --steps 10 --warmup 3 --rows 2000000 --ingest-chunks 0 ) > "$OUT/bench_g4.log" 2> "$OUT/bench_g4.err"
tail -c 1500 "$OUT/bench_g4.err"; head -c 1500 "$OUT/bench_g4.log"
