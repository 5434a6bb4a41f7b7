#!/bin/bash
# kernel trace of the search bench at the final code state (the command of scripts/profile_search.sh, first pass) + the encoder legs
cd /tmp && export TMPDIR=/tmp
ROOT="$GRAFT_REPO_ROOT"; OUT=$ROOT/gpurun_out/r5w; mkdir -p $OUT
BENCH="python $ROOT/bench.py --ingest-chunks 0 --bge-chunks 0 --no-cpu-baseline --shard-legs 0 --enc-like-rows 0 --text-docs 0 --precise-chunks 0"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH --steps 50 --warmup 10 --alt-steps 20 --side-steps 20 --small-steps 0 > "$ROOT/gpurun_out/r5w_bench_under_rocprof.json" 2> "$OUT/stats.log"
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); cp "$f" $ROOT/gpurun_out/r5w_bench_kernel_stats.csv; head -8 "$f" | cut -c1-140
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/enc" -- python $ROOT/scripts/gpu_encoder_prof.py l6 > /dev/null 2> "$OUT/enc.log"
f=$(find $OUT/enc -name "*kernel_stats.csv" | head -1); cp "$f" $ROOT/gpurun_out/r5w_encoder_kernel_stats.csv; head -6 "$f" | cut -c1-140
rm -rf $OUT
