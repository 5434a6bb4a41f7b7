#!/bin/bash
# r6n: the whole GPU suite + smoke
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r6n_tests.txt
cat gpurun_out/r6n_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/r6n_smoke.txt
