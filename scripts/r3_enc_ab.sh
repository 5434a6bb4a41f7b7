#!/bin/bash
# in-situ A/B of the encoder's kernel choices (same box, same process layout): chunks/s of scripts/gpu_encoder_perf.py
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd $ROOT
for rep in 1 2; do
for cfg in "1 0" "1 1" "1 4" "1 5" "2 0" "2 5" "1 2"; do
  set -- $cfg
  echo "TAIL=$1 GEMM_BIG=$2: $(MEMEX_HIP_TAIL=$1 MEMEX_HIP_GEMM_BIG=$2 timeout 300 python scripts/gpu_encoder_perf.py 2>/dev/null | grep -E "B=2048 S=512 ragged=False|H768" | sed 's/ragged=False: //; s/ chunks\/s wall.*gpu, / /; s/tokens.*//' | tr '\n' '|')"
done; done
