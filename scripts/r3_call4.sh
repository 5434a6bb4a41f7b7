#!/bin/bash
# round 3, GPU call 4: cold-load rate, whole GPU suite, then the profile suite (tag r3)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/c4; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
( time timeout 600 python -m pytest tests/test_persistence_gpu.py -m gpu -q -s ) > "$OUT/persist.log" 2>&1
grep -E "GB/s|passed|failed|Error" "$OUT/persist.log" | tail -8
( time timeout 2400 python -m pytest tests -m gpu -q ) > "$OUT/pytest.log" 2>&1
tail -8 "$OUT/pytest.log"
timeout 2400 bash scripts/profile_search.sh r3 > "$OUT/suite.log" 2>&1
tail -5 "$OUT/suite.log"
cat "$ROOT/gpurun_out/prof_r3/bench_default.json" | tail -c 1500
