#!/bin/bash
# r6m: sample size of the centred int8 copy (enc_like leg), MEMEX_HIP_DEBUG=sample_div=N
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
for dv in 16 32 64 128; do
  MEMEX_HIP_DEBUG=sample_div=$dv timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --ingest-chunks 0 --bge-chunks 0 --short-seqs 0 --precise-chunks 0 --text-docs 0 --shard-legs 0 --cfg2-segments 0 --alt-steps 0 --sides-out gpurun_out/r6m_sides_$dv.json > gpurun_out/r6m_bench_$dv.json 2> /dev/null
  python - $dv <<'P'
import json, sys
d=json.load(open(f'gpurun_out/r6m_bench_{sys.argv[1]}.json'))
e=d['sides']['enc_like_10M']
print('sample_div', sys.argv[1], 'headline', d['value'], 'cand', d['candidates_per_query'], '| enc_like', e['value'], 'ms', e['ms_per_step'], 'launch', e['ms_per_launch'], 'outside', e['ms_outside_collect_launch'], 'cand', e['candidates_per_query'], 'retry', e['retry_queries'])
P
done | tee gpurun_out/r6m_sample_div.txt
