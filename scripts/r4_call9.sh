#!/bin/bash
# round 4, call 9: attention_kernel<64> with its key loop software-pipelined: encoder tests, per-kernel averages, in-situ rate
mkdir -p gpurun_out
out=gpurun_out/r4_attn64.txt
: > $out
timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_cfg2_gpu.py -q -m gpu 2>&1 | tail -3 >> $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/st; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python $GRAFT_REPO_ROOT/scripts/gpu_encoder_prof.py bge > /dev/null 2>&1
python - >> $GRAFT_REPO_ROOT/$out <<'PY'
import csv, glob
f = glob.glob("/tmp/st/**/*_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "mx::" in r["Name"]: print(r["Name"].split("(")[0][:60], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY
cd $GRAFT_REPO_ROOT
for rep in 1 2; do timeout 300 python scripts/r4_enc_ab.py bge 6 2>&1 | grep -v amdgpu.ids >> $out; done
cat $out
