#!/bin/bash
# Package power and sclk (rocm-smi rows of scripts/power_sampler.py; the hwmon files can belong to another
# GPU of the host) while ablation variants of the fused-MLP microbenchmark run for ~5 s each.
#   bash scripts/r2_tail_power.sh 0 4 13      (builds build_ub/tail_ub_aN if missing)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/tail_power; rm -rf $OUT; mkdir -p $OUT
for a in "$@"; do
  [ -x $ROOT/build_ub/tail_ub_a$a ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMX_TAIL_ABLATE=$a -I $ROOT/memex_amd/csrc $ROOT/scripts/tail_ubench.hip $ROOT/memex_amd/csrc/encoder_tail.hip -o $ROOT/build_ub/tail_ub_a$a
  timeout 90 python $ROOT/scripts/power_sampler.py $OUT/a$a.log -- $ROOT/build_ub/tail_ub_a$a 131072 1536 ${REPS:-15000} > $OUT/a$a.txt 2>&1
  grep "^tail" $OUT/a$a.txt
  python - $OUT/a$a.log <<'PY'
import re, sys
rows = []
for ln in open(sys.argv[1]):
    if ln.startswith("R"):
        m = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", ln); p = re.search(r"Package Power \(W\): ([\d.]+)", ln)
        if m and p: rows.append((float(p.group(1)), int(m.group(1))))
busy = sorted(r for r in rows if r[0] > 900)
if busy:
    print("   under load (%d samples): power median %.0f W, sclk median %d MHz (min %d, max %d)" % (
        len(busy), busy[len(busy) // 2][0], sorted(b[1] for b in busy)[len(busy) // 2], min(b[1] for b in busy), max(b[1] for b in busy)))
PY
done
