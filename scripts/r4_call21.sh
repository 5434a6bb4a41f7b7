#!/bin/bash
# persistent attention grid with a length-sorted work list: tests, uniform kernel averages, ragged + uniform ingest rates
mkdir -p gpurun_out; out=$GRAFT_REPO_ROOT/gpurun_out/r4_attn_persist.txt; : > $out
timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_cfg2_gpu.py tests/test_pipeline_native_gpu.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error|assert" | head -8 >> $out
cd /tmp && export TMPDIR=/tmp
for m in bge l6; do
rm -rf /tmp/st; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python $GRAFT_REPO_ROOT/scripts/gpu_encoder_prof.py $m > /dev/null 2>&1
python - $m >> $out <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/st/**/*_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "attention" in r["Name"] or "attn_plan" in r["Name"]: print(sys.argv[1], r["Name"].split("(")[0][:44], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
done
cd $GRAFT_REPO_ROOT && timeout 600 python bench.py --steps 5 --warmup 2 --alt-steps 0 --side-steps 0 --small-steps 0 --shard-legs 0 --enc-like-rows 0 --no-cpu-baseline --bge-chunks 8192 --ingest-chunks 65536 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ingest', d['ingest']['value'], d['ingest']['roofline']['frac'], 'ragged', d['ingest']['ragged']['value'])
print('bge', d['ingest_bge_base']['value'], d['ingest_bge_base']['roofline']['frac'])
print('cfg2', d['cfg2']['embed_segments_per_s'])" >> $out
cat $out
