#!/bin/bash
# round 3, GPU call 8: per-half-tile certificate of the int8 copy: search tests, then sample size sweep
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/c8; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
( time timeout 1200 python -m pytest tests/test_search_gpu.py tests/test_compressed_gpu.py tests/test_sharded_gpu.py tests/test_cfg2_gpu.py -m gpu -q ) > "$OUT/pytest.log" 2>&1
grep -E "passed|failed" "$OUT/pytest.log" | tail -3; grep -E "^(FAILED|ERROR)|^E  " "$OUT/pytest.log" | head -30
DIVS="8 16 32" bash scripts/r3_i8_sample.sh
