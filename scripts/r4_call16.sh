#!/bin/bash
# d = 32 attention with two heads per stage: tests, kernel averages (pair / single, full / no key loop / key loop alone), end to end
mkdir -p gpurun_out; out=$GRAFT_REPO_ROOT/gpurun_out/r4_attn7.txt; : > $out
timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_cfg2_gpu.py tests/test_pipeline_native_gpu.py -q -m gpu -x 2>&1 | tail -3 >> $out
cd /tmp && export TMPDIR=/tmp
run() {  # model safe pair
rm -rf /tmp/st; MEMEX_HIP_ATTN_SAFE=$2 MEMEX_HIP_ATTN_PAIR=$3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python $GRAFT_REPO_ROOT/scripts/gpu_encoder_prof.py $1 > /dev/null 2>&1
python - $1 $2 $3 >> $out <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/st/**/*_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "attention" in r["Name"]: print("%s SAFE=%s PAIR=%s" % tuple(sys.argv[1:4]), r["Name"].split("(")[0][:44], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
}
for pair in 1 0; do for safe in 0 2; do run l6 $safe $pair; done; done; run bge 2 1
run bge 0 1
cd $GRAFT_REPO_ROOT && timeout 300 python scripts/r4_enc_ab.py both 6 2>&1 | grep -v amdgpu.ids >> $out
cat $out
