#!/bin/bash
# round 6: attention_x3_kernel with K / V rows fetched 2-4 key blocks ahead: parity tests, throughput, kernel time
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_encoder_gpu.py tests/test_pipeline_native_gpu.py -m gpu -x -q > gpurun_out/r6bb_tests.txt 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r6bb_tests.txt
timeout 900 python scripts/gpu_encoder_precise.py 2>&1 | grep "chunks/s" | tee gpurun_out/r6bb_precise.txt
for m in bge l6; do
  rm -rf /tmp/mp; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mp -- python $GRAFT_REPO_ROOT/scripts/gpu_encoder_prof.py $m mixed > /tmp/mp.log 2>&1)
  f=$(find /tmp/mp -name "*_kernel_stats.csv" | head -1)
  echo "== $m mixed"
  python -c "
import csv
rows=[r for r in csv.DictReader(open('$f')) if 'mx::' in r['Name']]
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:6]: print('%-70s calls %5s avg %9.1f us  %5.1f %%' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1000, 100*float(r['TotalDurationNs'])/tot))
"
done 2>&1 | tee gpurun_out/r6bb_mixed_kernels.txt
