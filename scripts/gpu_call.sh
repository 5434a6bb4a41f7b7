#!/bin/bash
# full check at the end of the session: GPU tests, smoke, default bench line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|rror|FAILED" | tail -8; echo "pytest rc=${PIPESTATUS[0]}" ) > gpurun_out/r5ab_check.txt 2>&1
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids >> gpurun_out/r5ab_check.txt
timeout 900 python bench.py > gpurun_out/r5ab_bench.json 2> gpurun_out/r5ab_bench.err
cat gpurun_out/r5ab_check.txt; head -c 300 gpurun_out/r5ab_bench.json
