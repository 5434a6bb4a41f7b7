#!/bin/bash
# Host completion wait of a search batch: polling the kernel's completion word (MEMEX_HIP_SPIN=1, what bench.py sets) against the
# library's default, a sleeping wait, on an idle host and with every core kept busy by a hog.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
run() { python $ROOT/scripts/r2_step_jitter.py | head -1; }
echo "idle host, spin:  $(MEMEX_HIP_SPIN=1 run)"
echo "idle host, sleep: $(run)"
N=$(nproc)
for i in $(seq 1 $N); do ( while :; do :; done ) & done
sleep 1
echo "busy host ($N hogs), spin:  $(MEMEX_HIP_SPIN=1 run)"
echo "busy host ($N hogs), sleep: $(run)"
kill $(jobs -p) 2>/dev/null
wait 2>/dev/null
