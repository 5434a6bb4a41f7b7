#!/bin/bash
# round 4, call 6: the default bench line on the round-4 build (new legs: shard proxies, encoder-like 10M, bge CPU baseline,
# settle phase), then the N>1 wiring checks on one device: both forms, and each form's fallback to the other
mkdir -p gpurun_out
python bench.py > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err
echo "rc=$?"; tail -c 600 gpurun_out/r4_bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_bench_default.json').read().strip().splitlines()[-1])
keep={k:d[k] for k in ('value','ms_per_step','settle_steps','candidates_per_query','ms_outside_collect_launch')}
print(keep, d['roofline']['frac'], d['roofline']['ms_per_launch'])
for k in ('host_api','clustered','anisotropic','cfg4_shard_10Mx768','shard_1p25Mx384','shard_1p25Mx768','enc_like_10M'):
    v=d.get(k)
    if v: print(k, {x:v.get(x) for x in ('value','ms_per_step','candidates_per_query','retry_queries','fallback_queries','scan','filter_demotions','ms_outside_collect_launch','predicted_n8_qps','scan_before_first_batch','first_batch')})
print('cfg2', d.get('cfg2'))
for k in ('ingest','ingest_bge_base'):
    v=d.get(k); print(k, v['value'], v['gpu_only_chunks_per_s'], v['roofline']['frac'], v.get('cpu_baseline'))
print('cpu', {k:v for k,v in d['cpu_baseline'].items() if k!='hnsw'})
PY
export MEMEX_BENCH_ONE_DEVICE=1
COMMON="--gpus 4 --rows 2000000 --steps 5 --warmup 2 --side-steps 0 --alt-steps 0 --no-cpu-baseline --ingest-chunks 0 --bge-chunks 0 --cfg2-segments 0 --min-seconds 0.1"
for mode in plain torchrun; do for fail in none in-library per-process; do
  echo "=== $mode fail=$fail"
  if [ $fail != none ]; then export MEMEX_BENCH_TEST_FAIL=$fail; else unset MEMEX_BENCH_TEST_FAIL; fi
  if [ $mode = plain ]; then timeout 600 python bench.py $COMMON 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('value','n_gpus','exchange','fallback_from','error','merged_lists_ok','ids_equal_exact_path')}, d['config']['parallelism'])"
  else timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29611 bench.py $COMMON 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('value','n_gpus','exchange','fallback_from','error','merged_lists_ok','ids_equal_exact_path')}, d['config']['parallelism'])"
  fi
done; done
