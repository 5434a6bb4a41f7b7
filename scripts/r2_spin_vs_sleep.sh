#!/bin/bash
# Host completion wait of a search batch: spinning on the kernel's completion word (default) against sleeping
# in hipStreamSynchronize (MEMEX_HIP_NO_SPIN=1), on an idle host and with every core kept busy by a hog.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
run() { python $ROOT/scripts/r2_step_jitter.py | head -1; }
echo "idle host, spin:  $(run)"
echo "idle host, sleep: $(MEMEX_HIP_NO_SPIN=1 run)"
N=$(nproc)
for i in $(seq 1 $N); do ( while :; do :; done ) & done
sleep 1
echo "busy host ($N hogs), spin:  $(run)"
echo "busy host ($N hogs), sleep: $(MEMEX_HIP_NO_SPIN=1 run)"
kill $(jobs -p) 2>/dev/null
wait 2>/dev/null
