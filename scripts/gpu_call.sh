#!/bin/bash
mkdir -p gpurun_out
timeout 300 python -c "
import json, bench
print(json.dumps(bench.precise_ingest_leg(16384, 4096), indent=1))
" > gpurun_out/r5_precise_leg.txt 2>&1
cat gpurun_out/r5_precise_leg.txt
