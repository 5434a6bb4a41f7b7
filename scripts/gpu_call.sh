#!/bin/bash
# the GPU call of the moment (see scripts/README.md): full GPU suite + smoke
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > /tmp/pt.log 2>&1
echo "pytest rc=$?" > gpurun_out/r5_final_check3.txt
grep -E "passed|failed|Error" /tmp/pt.log | tail -4 >> gpurun_out/r5_final_check3.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> gpurun_out/r5_final_check3.txt
cat gpurun_out/r5_final_check3.txt
