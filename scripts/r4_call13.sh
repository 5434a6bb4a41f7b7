#!/bin/bash
# streaming attention kernel: tests, then kernel averages (full / no key loop) and the end-to-end A/B
mkdir -p gpurun_out; out=$GRAFT_REPO_ROOT/gpurun_out/r4_attn4.txt; : > $out
timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_cfg2_gpu.py tests/test_pipeline_native_gpu.py -q -m gpu -x 2>&1 | tail -15 >> $out
cd /tmp && export TMPDIR=/tmp
run() {  # model safe hpw
rm -rf /tmp/st; MEMEX_HIP_ATTN_SAFE=$2 MEMEX_HIP_ATTN_HPW=$3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python $GRAFT_REPO_ROOT/scripts/gpu_encoder_prof.py $1 > /dev/null 2>&1
python - $1 $2 $3 >> $out <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/st/**/*_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "attention" in r["Name"]: print("%s SAFE=%s HPW=%s" % tuple(sys.argv[1:4]), r["Name"].split("(")[0][:40], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
}
for m in bge l6; do for safe in 0 1 2; do run $m $safe 0; done; done
for hpw in 1 3 6 12; do run bge 0 $hpw; run l6 0 $hpw; done
cd $GRAFT_REPO_ROOT && timeout 300 python scripts/r4_enc_ab.py both 6 2>&1 | grep -v amdgpu.ids >> $out
cat $out
