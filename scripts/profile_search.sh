#!/bin/bash
# Profiles the search bench on the GPU box: --kernel-trace --stats passes (10M x 384 with its side legs;
# 10M x 768 alone), then separate --pmc passes (never combined with other trace domains), package
# power / clock logs of the scan kernel, of the encoder's tail kernel and of their ablations, the encoder
# kernels and the default bench
# line; scripts/profile_reduce.py writes the summaries that get copied into profiles/.
# Usage: scripts/profile_search.sh [tag]
set -u
TAG=${1:-r4}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# (the 1.25M-row, encoder-like and cfg2 legs launch the same kernels at other sizes -- cfg2 runs scan8_kernel<3,1,1> on 100k rows since
# its copy stays int8 (round 6): off under the profiler, like --small-steps 0)
BENCH="python $ROOT/bench.py --ingest-chunks 0 --bge-chunks 0 --short-seqs 0 --no-cpu-baseline --shard-legs 0 --enc-like-rows 0 --cfg2-segments 0 --text-docs 0 --precise-chunks 0 --sides-out $OUT/sides_scratch.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH --steps 50 --warmup 10 --alt-steps 20 --side-steps 20 --small-steps 0 > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats768" -- $BENCH --dim 768 --steps 30 --warmup 5 --alt-steps 0 --side-steps 0 > "$OUT/bench768_under_rocprof.json" 2> "$OUT/stats768.log"
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"; do
  D=$OUT/pmc_$(echo $C | tr ' ' '_' | cut -c1-60)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$D" -- $BENCH --steps 6 --warmup 2 --alt-steps 6 --side-steps 0 > "$D.json" 2> "$D.log"
  D=$OUT/pmc768_$(echo $C | tr ' ' '_' | cut -c1-60)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$D" -- $BENCH --dim 768 --steps 6 --warmup 2 --alt-steps 0 --side-steps 0 > "$D.json" 2> "$D.log"
done
$BENCH --sides-out "$OUT/bench_sides.json" > "$OUT/bench.json" 2> "$OUT/bench.log"
# the centred int8 copy (round 6): the enc_like leg alone under the kernel trace
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/enc_like_stats" -- python $ROOT/scripts/gpu_enc_like.py 10000000 30 > "$OUT/enc_like.json" 2> "$OUT/enc_like_stats.log"
# package power and shader clock while the scan kernel (and its ablations: 1 = DMA stream only,
# 4 = fragment reads + MFMA without the DMA) runs back to back for >= 5 s
if [ -x $ROOT/build_ub/scan8_ub_0 ]; then   # the int8-copy scan on signed Gaussian-like bytes, no records
  python $ROOT/scripts/power_sampler.py "$OUT/power_scan8.log" -- $ROOT/build_ub/scan8_ub_0 10000000 384 6000 > /dev/null 2>&1
  python $ROOT/scripts/power_sampler.py "$OUT/power_scan8_768.log" -- $ROOT/build_ub/scan8_ub_0 10000000 768 3000 > /dev/null 2>&1
fi
if [ -x $ROOT/build_ub/scan16_ub_0 ]; then
  for V in 0 1 4; do
    python $ROOT/scripts/power_sampler.py "$OUT/power_scan16_ablate$V.log" -- $ROOT/build_ub/scan16_ub_$V 10000000 384 3000 > /dev/null 2>&1
  done
  python $ROOT/scripts/power_sampler.py "$OUT/power_scan16_768.log" -- $ROOT/build_ub/scan16_ub_0 10000000 768 1500 > /dev/null 2>&1
fi
# the encoder's layer-tail kernel alone (131072 tokens, ffn 1536) and two ablations (4 = no GELU arithmetic,
# 13 = no weight loads, no LDS fragment reads, no GELU): time per launch + package power / clock over >= 5 s
: > "$OUT/tail_ubench.txt"
for V in 0 4 13; do
  [ -x $ROOT/build_ub/tail_ub_a$V ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DMX_TAIL_ABLATE=$V -I $ROOT/memex_amd/csrc -I $ROOT/scripts $ROOT/scripts/tail_ubench.hip $ROOT/memex_amd/csrc/encoder_tail.hip $ROOT/scripts/encoder_tail2.hip -o $ROOT/build_ub/tail_ub_a$V
  timeout 120 python $ROOT/scripts/power_sampler.py "$OUT/power_tail_ablate$V.log" -- $ROOT/build_ub/tail_ub_a$V 131072 1536 12000 2>&1 | grep "^tail" >> "$OUT/tail_ubench.txt"
done
# encoder kernels (MiniLM-L6 shape, 2048 x 512-token chunks) and the default bench line (all legs)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/enc_stats" -- python $ROOT/scripts/gpu_encoder_prof.py l6 > /dev/null 2> "$OUT/enc_stats.log"
# ... and the bge-base shape (1024 chunks): pgemm_kernel + ln_rows_kernel + attention_kernel<64>; HBM traffic of its kernels (PMC passes of their own)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/enc_bge_stats" -- python $ROOT/scripts/gpu_encoder_prof.py bge > /dev/null 2> "$OUT/enc_bge_stats.log"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/encpmc_$C" -- python $ROOT/scripts/gpu_encoder_prof.py bge > /dev/null 2> "$OUT/encpmc_$C.log"
done
python $ROOT/scripts/profile_encoder_traffic.py "$OUT" > "$OUT/encoder_bge_traffic.txt" 2>&1
# effective shader clock and MFMA-busy fraction of every encoder kernel (PMC pass of its own)
timeout 300 bash $ROOT/scripts/profile_encoder_clock.sh > "$OUT/encoder_clock_mfma.txt" 2>&1
python $ROOT/bench.py --sides-out "$OUT/bench_default_sides.json" > "$OUT/bench_default.json" 2> "$OUT/bench_default.log"
python "$ROOT/scripts/profile_reduce.py" "$OUT" "$TAG"
# the raw rocprofv3 directories are large (gpurun brings back at most 64 MiB): keep the summaries and the logs
rm -rf "$OUT"/enc_like_stats "$OUT"/stats "$OUT"/stats768 "$OUT"/enc_stats "$OUT"/enc_bge_stats "$OUT"/encpmc_*/ "$OUT"/pmc_*/ "$OUT"/pmc768_*/
