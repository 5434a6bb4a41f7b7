#!/bin/bash
# runs the scan16 microbenchmark binaries (build_ub/)
cd "$(dirname "$0")/.."
for b in build_ub/scan16_ub_*; do $b 10000000 384 20; done
