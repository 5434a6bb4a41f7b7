#!/bin/bash
# round 3, GPU call 3: whole GPU suite, GEMM / tail ubench after the tile and tail2 changes, bench.py + kernel stats
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/c3; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
( time timeout 2400 python -m pytest tests -m gpu -q ) > "$OUT/pytest.log" 2>&1
tail -12 "$OUT/pytest.log"
{ timeout 100 $ROOT/build_ub/tail_ub_tr 131072 1536 500 1 0 | grep -E "^tail|vs tail"; timeout 200 $ROOT/build_ub/gemm_ub 131072 384 1536 100 | grep -E "^vt|^qk"; timeout 200 $ROOT/build_ub/gemm_ub 131072 768 3072 40 | grep -E "^vt"; } > "$OUT/ub.log" 2>&1
cat "$OUT/ub.log"
timeout 300 python scripts/gpu_encoder_perf.py > "$OUT/enc_perf.log" 2>&1; tail -6 "$OUT/enc_perf.log"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/enc_stats" -- python $ROOT/scripts/gpu_encoder_prof.py l6 > /dev/null 2> "$OUT/enc_stats.log"
python - "$OUT/enc_stats" <<'PY'
import csv, glob, sys, os
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True)
for r in list(csv.DictReader(open(f[0])))[:9]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']}%")
PY
cd "$ROOT"
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > "$OUT/bench.log" 2> "$OUT/bench.err"
tail -c 600 "$OUT/bench.err"
python - "$OUT/bench.log" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        d = json.loads(ln)
        print("value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "outside", d["ms_outside_collect_launch"])
        print("ingest", d["ingest"]["value"], d["ingest"]["roofline"]["frac"], "bge", d["ingest_bge_base"]["value"], d["ingest_bge_base"]["roofline"]["frac"])
        print("cfg2", {k: d["cfg2"][k] for k in ("embed_segments_per_s", "embed_mfma_frac", "search_value")})
PY
