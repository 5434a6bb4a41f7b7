#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r5_text_ingest_profile.txt
: > $O
timeout 300 python scripts/gpu_text_ingest_profile.py 50 >> $O 2>&1
timeout 600 python -c "
import json, bench
r = bench.text_ingest_leg(200, cpu_too=False); print(json.dumps({k: r[k] for k in ('value','windows_per_s','text_MBps','seconds','errors','query_finds_its_window')}))
r = bench.text_ingest_leg(200, workers=16, cpu_too=False); print('16 workers', json.dumps({k: r[k] for k in ('value','windows_per_s','text_MBps','seconds','errors','query_finds_its_window')}))
" >> $O 2>&1
cat $O
