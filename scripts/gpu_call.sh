#!/bin/bash
# r6h: the default bench line with the restructured record (every roofline claim nested under `roofline`)
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6h_bench.json 2> gpurun_out/r6h_bench.err ) 2> gpurun_out/r6h_time.txt
tail -c 3000 gpurun_out/r6h_bench.json; echo; tail -3 gpurun_out/r6h_time.txt; grep -v "side legs" gpurun_out/r6h_bench.err | tail -5; wc -c gpurun_out/r6h_bench.json
