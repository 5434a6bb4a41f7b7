#!/bin/bash
# round 6: full GPU suite + smoke + default bench at HEAD
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r6f_tests.txt 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r6f_tests.txt
grep -E "passed|failed" gpurun_out/r6f_tests.txt | tail -2
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r6f_smoke.txt 2>&1; tail -2 gpurun_out/r6f_smoke.txt
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 --sides-out gpurun_out/r6f_bench_sides.json > gpurun_out/r6f_bench.json 2> gpurun_out/r6f_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
r = json.loads(open('gpurun_out/r6f_bench.json').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'], r['roofline']['frac'])
for k in ('encoder_minilm', 'encoder_bge'):
    print(k, r['roofline'][k]['chunks_per_s'], r['roofline'][k]['frac'])
for m, v in r['roofline']['encoder_split_modes'].items():
    print(m, {k: (round(x['value']) if isinstance(x, dict) and 'value' in x else x) for k, x in v.items() if k != 'score_error_note'})
print(json.dumps(r['sides'].get('enc_like_10M')))
PY
