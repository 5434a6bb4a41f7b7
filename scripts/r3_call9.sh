#!/bin/bash
# round 3, GPU call 9: the profile suite (tag r3) with the int8 filter copy as the default
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/c9; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
timeout 3000 bash scripts/profile_search.sh r3 > "$OUT/suite.log" 2>&1
tail -8 "$OUT/suite.log" | cut -c1-500
