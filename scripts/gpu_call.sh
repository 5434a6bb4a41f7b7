#!/bin/bash
# round 6: full GPU suite + smoke at HEAD (the last one of the round)
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r6f_tests.txt 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r6f_tests.txt
grep -E "passed|failed" gpurun_out/r6f_tests.txt | tail -2
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r6f_smoke.txt 2>&1; tail -2 gpurun_out/r6f_smoke.txt
