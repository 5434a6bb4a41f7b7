#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r5_text_ingest_pipelined.txt
: > $O
timeout 900 python -m pytest tests/test_pipeline_native_gpu.py tests/test_pretrained.py tests/test_encoder_gpu.py -m gpu -x -q -k "embedder or pipeline or pretrained or cfg1" 2>&1 | tail -3 >> $O
timeout 600 python -c "
import json, bench
for w in (1, 5, 5, 16):
    r = bench.text_ingest_leg(200, workers=w, cpu_too=False); print(w, 'workers', json.dumps({k: r[k] for k in ('value','windows_per_s','text_MBps','seconds','errors','query_finds_its_window')}))
" >> $O 2>&1
cat $O
