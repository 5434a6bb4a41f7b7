#!/bin/bash
# round 3: int8 against bf16 filter copy by row width (Gaussian rows, B = 256, k = 10); the int8 scan at several
# sample sizes (MEMEX_HIP_DEBUG sample_div=N; 0 = the library's choice)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/dims
{ echo "== filter bf16"; MEMEX_HIP_FILTER=bf16 timeout 900 python scripts/gpu_search_shapes.py dims 2>&1 | grep "^n="
  for DIV in ${DIVS:-0 4 8 16 32}; do
    echo "== filter i8 sample_div $DIV"
    if [ "$DIV" = 0 ]; then MEMEX_HIP_FILTER=i8 timeout 900 python scripts/gpu_search_shapes.py dims 2>&1 | grep "^n="
    else MEMEX_HIP_DEBUG=sample_div=$DIV MEMEX_HIP_FILTER=i8 timeout 900 python scripts/gpu_search_shapes.py dims 2>&1 | grep "^n="; fi
  done; } | tee gpurun_out/dims/dims.log
