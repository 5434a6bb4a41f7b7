#!/bin/bash
# r6i: centred int8 copy -- the search tests, then the enc_like / cfg2 legs
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_centred_gpu.py tests/test_search_gpu.py tests/test_random_ops_gpu.py tests/test_cfg2_gpu.py -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r6i_tests.txt
cat gpurun_out/r6i_tests.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --ingest-chunks 0 --bge-chunks 0 --short-seqs 0 --precise-chunks 0 --text-docs 0 --shard-legs 0 --sides-out gpurun_out/r6i_sides.json > gpurun_out/r6i_bench.json 2> gpurun_out/r6i_bench.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r6i_bench.json'))
print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'])
for k in ('enc_like_10M','cfg2','clustered','anisotropic'):
    print(k, d['sides'].get(k))
P
tail -3 gpurun_out/r6i_bench.err
