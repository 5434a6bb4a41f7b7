#!/bin/bash
# round 3: int8 against bf16 filter copy by row width (Gaussian rows, B = 256, k = 10)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/dims
for F in bf16 i8; do for DIV in ${DIVS:-0 4}; do
  echo "== filter $F sample_div ${DIV}"
  if [ "$DIV" = 0 ]; then MEMEX_HIP_FILTER=$F timeout 900 python scripts/gpu_search_shapes.py dims 2>&1 | grep "^n="; else
  [ $F = i8 ] && MEMEX_HIP_SAMPLE_DIV=$DIV MEMEX_HIP_FILTER=$F timeout 900 python scripts/gpu_search_shapes.py dims 2>&1 | grep "^n="; fi
done; done | tee gpurun_out/dims/dims.log
