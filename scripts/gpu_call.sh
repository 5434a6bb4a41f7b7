#!/bin/bash
# round 6: full GPU suite + smoke + default bench with the centred int8 copy on accumulator initial values
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r6p_tests.txt 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r6p_tests.txt
grep -E "passed|failed" gpurun_out/r6p_tests.txt | tail -2
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r6p_smoke.txt 2>&1; tail -2 gpurun_out/r6p_smoke.txt
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 --sides-out gpurun_out/r6p_bench_sides.json > gpurun_out/r6p_bench.json 2> gpurun_out/r6p_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
r = json.loads(open('gpurun_out/r6p_bench.json').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'], r['roofline']['frac'])
s = r['sides']
for k in ('enc_like_10M', 'cfg2_1Mx768', 'cfg2'):
    if k in s: print(k, json.dumps(s[k])[:600])
print([k for k in s])
PY
