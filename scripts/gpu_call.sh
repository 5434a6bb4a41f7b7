#!/bin/bash
# r6p: kernel stats of the headline command without the legs that launch the same kernel at other sizes
cd /tmp && export TMPDIR=/tmp
ROOT="$GRAFT_REPO_ROOT"; OUT=$ROOT/gpurun_out/r6p; rm -rf $OUT; mkdir -p $OUT
BENCH="python $ROOT/bench.py --ingest-chunks 0 --bge-chunks 0 --short-seqs 0 --no-cpu-baseline --shard-legs 0 --enc-like-rows 0 --cfg2-segments 0 --text-docs 0 --precise-chunks 0 --sides-out $OUT/sides.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH --steps 50 --warmup 10 --alt-steps 20 --side-steps 20 --small-steps 0 > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log"
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); cp "$f" $ROOT/gpurun_out/r6p_bench_kernel_stats.csv
head -8 $ROOT/gpurun_out/r6p_bench_kernel_stats.csv | cut -c1-160
tail -c 600 "$OUT/bench_under_rocprof.json" > /dev/null; cp "$OUT/bench_under_rocprof.json" $ROOT/gpurun_out/r6p_bench_under_rocprof.json; rm -rf $OUT/stats
