#!/bin/bash
# full GPU suite (no -x: every failure is listed), smoke
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|rror|FAILED" | tail -12; echo "pytest rc=${PIPESTATUS[0]}" ) > gpurun_out/r5v_check.txt 2>&1
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids >> gpurun_out/r5v_check.txt
cat gpurun_out/r5v_check.txt
