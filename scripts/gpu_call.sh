#!/bin/bash
# full check with the short-sequence attention in place: GPU tests, smoke, default bench line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5; echo "pytest rc=${PIPESTATUS[0]}" ) > gpurun_out/r5u_check.txt 2>&1
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids >> gpurun_out/r5u_check.txt
timeout 900 python bench.py > gpurun_out/r5u_bench.json 2> gpurun_out/r5u_bench.err
timeout 300 python scripts/gpu_encoder_perf.py short 2>&1 | grep chunks > gpurun_out/r5u_perf.txt
cat gpurun_out/r5u_check.txt gpurun_out/r5u_perf.txt; head -c 300 gpurun_out/r5u_bench.json
