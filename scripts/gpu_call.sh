#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_encoder_gpu.py tests/test_pipeline_native_gpu.py tests/test_pretrained.py tests/test_cfg2_gpu.py -m gpu -x -q > /tmp/pt.log 2>&1
echo "pytest rc=$?" > gpurun_out/r5_enc_tests.txt
grep -E "passed|failed|Error|assert" /tmp/pt.log | tail -15 >> gpurun_out/r5_enc_tests.txt
timeout 300 python scripts/gpu_doc_pass_prof.py 8 128 2>&1 | grep "per encode" >> gpurun_out/r5_enc_tests.txt
timeout 300 python scripts/gpu_doc_pass_prof.py 16 128 2>&1 | grep "per encode" >> gpurun_out/r5_enc_tests.txt
timeout 600 python -c "
import json, bench
for w in (1, 5):
    r = bench.text_ingest_leg(200, workers=w, cpu_too=False); print(w, 'workers', json.dumps({k: r[k] for k in ('value','windows_per_s','text_MBps','seconds','errors','query_finds_its_window')}))
" 2>&1 | grep workers >> gpurun_out/r5_enc_tests.txt
cat gpurun_out/r5_enc_tests.txt
