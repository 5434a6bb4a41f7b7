#!/bin/bash
mkdir -p gpurun_out
timeout 2400 bash scripts/profile_search.sh r5 > gpurun_out/r5_profile_run.log 2>&1
echo "rc=$?" >> gpurun_out/r5_profile_run.log
tail -5 gpurun_out/r5_profile_run.log
ls gpurun_out/prof_r5 | head -50
