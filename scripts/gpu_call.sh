#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r5_splitk_768_sweep.txt
: > $O
timeout 300 python scripts/gpu_query_latency.py bge 16x128,24x128,31x128,40x128 >> $O 2>&1
echo "-- MEMEX_HIP_SPLITK=0" >> $O
MEMEX_HIP_SPLITK=0 timeout 300 python scripts/gpu_query_latency.py bge 16x128,24x128,31x128,40x128 >> $O 2>&1
cat $O
