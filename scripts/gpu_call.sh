#!/bin/bash
# default bench line with its per-stage wall times
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( time timeout 1200 python bench.py > gpurun_out/r5i_bench.json 2> gpurun_out/r5i_bench.err ) 2> gpurun_out/r5i_time.txt
cat gpurun_out/r5i_time.txt; python -c "
import json; d=json.loads(open('gpurun_out/r5i_bench.json').read().strip().splitlines()[-1]); print(json.dumps(d['wall_s'], indent=0)); print(d['value'], d['ingest']['value'], d['ingest_bge_base']['value'])"
