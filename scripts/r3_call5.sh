#!/bin/bash
# round 3, GPU call 5: wide rows (768 < dim <= 1536) through scan16w_kernel: parity, then throughput
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/c5; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
( time timeout 900 python -m pytest tests/test_search_gpu.py tests/test_compressed_gpu.py -m gpu -q -x ) > "$OUT/pytest.log" 2>&1
tail -15 "$OUT/pytest.log"
timeout 600 python scripts/gpu_search_shapes.py wide > "$OUT/wide.log" 2>&1
cat "$OUT/wide.log" | tail -8
