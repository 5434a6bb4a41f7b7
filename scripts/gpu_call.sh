#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r5_text_ingest.txt
: > $O
nproc >> $O
timeout 900 python -m pytest tests/test_cpp_host.py tests/test_pipeline_native_gpu.py tests/test_pretrained.py -m gpu -x -q 2>&1 | tail -5 >> $O
timeout 600 python -c "
import json, bench
print(json.dumps(bench.text_ingest_leg(200), indent=1))
print(json.dumps(bench.text_ingest_leg(200, workers=32, cpu_too=False), indent=1))
" >> $O 2>&1
cat $O
