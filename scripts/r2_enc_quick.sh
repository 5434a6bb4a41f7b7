#!/bin/bash
# encoder: parity tests, per-kernel averages (rocprofv3 --stats) and throughput
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/${1:-enc}
rm -rf "$OUT"; mkdir -p "$OUT"
cd $ROOT
python -m pytest tests/test_encoder_gpu.py tests/test_pipeline_native_gpu.py -x -q 2>&1 | tail -8 > "$OUT/pytest.log"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python $ROOT/scripts/gpu_encoder_prof.py l6 > /dev/null 2> "$OUT/stats.log"
python - "$OUT/stats" <<'PY'
import csv, glob, sys, os
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True)
for r in list(csv.DictReader(open(f[0])))[:9]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']}%")
PY
python $ROOT/scripts/gpu_encoder_perf.py 2>&1 | tail -8
tail -3 "$OUT/pytest.log"
