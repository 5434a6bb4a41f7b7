#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/r5_scan8_shape_probe.txt
for rep in 1 2; do for v in 0 1; do
  timeout 120 build_ub/scan8_shape_$v 10000000 384 1500 >> gpurun_out/r5_scan8_shape_probe.txt 2>&1
done; done
for v in 0 1; do timeout 120 build_ub/scan8_shape_$v 10000000 768 800 >> gpurun_out/r5_scan8_shape_probe.txt 2>&1; done
cat gpurun_out/r5_scan8_shape_probe.txt
