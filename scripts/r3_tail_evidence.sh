#!/bin/bash
# round 3: the evidence behind DESIGN.md section 4 "Round 3": phase trace of tail_kernel and tail2_kernel
# (build_ub/tail_ub_tr = scripts/tail_ubench.hip with -DMX_TAIL_TRACE=1 -DMX_TAIL2_TRACE=1), tail2's ablations
# (build_ub/t2_abN = -DMX_TAIL2_ABLATE=N: 1 no wait/barrier, 2 no DMA, 4 no GELU, 8 no fragment reads), and the
# in-situ A/B of the encoder's kernel choices (scripts/r3_enc_ab.sh)
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd $ROOT
OUT=$ROOT/gpurun_out/ev; rm -rf $OUT; mkdir -p $OUT
timeout 120 build_ub/tail_ub_tr 131072 1536 1500 1 0 > $OUT/r3_tail_trace.txt 2>&1
{ for v in 1 2 3 4 8 15; do [ -x build_ub/t2_ab$v ] && timeout 120 build_ub/t2_ab$v 131072 1536 1000 1 0 2>&1 | grep -E "^tail2 (m=|blocks +0)" | sed "s/^/ablate=$v: /"; done
  timeout 120 build_ub/tail_ub_tr 131072 1536 1000 1 0 2>&1 | grep -E "^tail2 (m=|blocks +0)" | sed "s/^/ablate=0: /"; } > $OUT/r3_tail2_ablations.txt
timeout 1500 bash scripts/r3_enc_ab.sh > $OUT/r3_encoder_ab.txt 2>&1
tail -20 $OUT/r3_tail2_ablations.txt; cat $OUT/r3_encoder_ab.txt
