#!/bin/bash
# full check at HEAD: GPU tests, smoke, default bench line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5; echo "pytest rc=${PIPESTATUS[0]}" ) > gpurun_out/r5c_check.txt 2>&1
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/r5c_check.txt 2>&1
timeout 600 python bench.py > gpurun_out/r5c_bench.json 2> gpurun_out/r5c_bench.err
tail -3 gpurun_out/r5c_check.txt; head -c 600 gpurun_out/r5c_bench.json
