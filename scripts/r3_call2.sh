#!/bin/bash
# round 3, GPU call 2: tail2_kernel (first run: correctness vs tail_kernel, time, phase trace), GPU tests touched this
# round, GEMM tile configurations (ubench), encoder throughput
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/c2; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
{
  timeout 120 $ROOT/build_ub/tail_ub_tr 131072 1536 1000 1 0
  timeout 60 $ROOT/build_ub/tail_ub 32768 1536 300 1 0 | grep -E "^tail"
  timeout 60 $ROOT/build_ub/tail_ub 131072 768 300 1 0 | grep -E "^tail"
} > "$OUT/tail.log" 2>&1
cat "$OUT/tail.log"
( time timeout 1500 python -m pytest tests/test_encoder_gpu.py tests/test_cfg2_gpu.py tests/test_sharded_gpu.py tests/test_persistence_gpu.py tests/test_cpp_host.py tests/test_pipeline_native_gpu.py -m gpu -q ) > "$OUT/pytest.log" 2>&1
tail -15 "$OUT/pytest.log"
{ timeout 300 $ROOT/build_ub/gemm_ub 131072 768 3072 60; timeout 200 $ROOT/build_ub/gemm_ub 131072 384 1536 100; } > "$OUT/gemm.log" 2>&1
cat "$OUT/gemm.log"
timeout 300 python scripts/gpu_encoder_perf.py > "$OUT/enc_perf.log" 2>&1; tail -8 "$OUT/enc_perf.log"
MEMEX_HIP_TAIL=1 timeout 300 python scripts/gpu_encoder_perf.py > "$OUT/enc_perf_tail1.log" 2>&1; tail -8 "$OUT/enc_perf_tail1.log"
