#!/bin/bash
# polynomial GELU: encoder tests, the GEMM / tail microbenchmarks, end-to-end A/B
mkdir -p gpurun_out; out=$GRAFT_REPO_ROOT/gpurun_out/r4_gelu.txt; : > $out
timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_cfg2_gpu.py tests/test_pipeline_native_gpu.py -q -m gpu -x 2>&1 | tail -3 >> $out
./build_ub/gemm_ub_p0 131072 768 3072 50 | grep "ffn1" >> $out
./build_ub/tail_ub_a0 131072 1536 300 2>&1 | grep "^tail " >> $out
timeout 300 python scripts/r4_enc_ab.py both 6 2>&1 | grep -v amdgpu.ids >> $out
cat $out
