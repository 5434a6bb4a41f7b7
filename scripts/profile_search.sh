#!/bin/bash
# Profiles the search bench on the GPU box: one --kernel-trace --stats pass, then separate --pmc
# passes (never combined with other trace domains), then scripts/profile_reduce.py writes the
# summaries that get copied into profiles/.  Usage: scripts/profile_search.sh [tag]
set -u
TAG=${1:-r1}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --ingest-chunks 0 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH --steps 50 --warmup 10 --alt-steps 20 > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log"
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  D=$OUT/pmc_$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$D" -- $BENCH --steps 6 --warmup 2 --alt-steps 6 > "$D.json" 2> "$D.log"
done
$BENCH > "$OUT/bench.json" 2> "$OUT/bench.log"
# encoder kernels (MiniLM-L6 shape, 2048 x 512-token chunks) and the default bench line (all legs)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/enc_stats" -- python $ROOT/scripts/gpu_encoder_prof.py l6 > /dev/null 2> "$OUT/enc_stats.log"
python $ROOT/bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.log"
python "$ROOT/scripts/profile_reduce.py" "$OUT" "$TAG"
