#!/bin/bash
mkdir -p gpurun_out; out=gpurun_out/r4_attn2.txt; : > $out
timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_cfg2_gpu.py tests/test_pipeline_native_gpu.py -q -m gpu 2>&1 | tail -3 >> $out
bash scripts/r4_call10.sh > /dev/null 2>&1; cat gpurun_out/r4_attn_staging.txt >> $out
timeout 300 python scripts/r4_enc_ab.py both 6 2>&1 | grep -v amdgpu.ids >> $out
cat $out
