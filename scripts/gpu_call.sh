#!/bin/bash
# round-5 record: the full GPU suite, then the profile recipe (kernel stats, PMC traffic, power logs, default bench line)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r5_tests_full.txt
cat gpurun_out/r5_tests_full.txt
timeout 3000 bash scripts/profile_search.sh r5 > gpurun_out/r5_profile.log 2>&1
tail -5 gpurun_out/r5_profile.log
ls gpurun_out/prof_r5 | head -50
