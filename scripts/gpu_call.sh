#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
bash scripts/r6_step_timeline.sh 2>&1 | tee gpurun_out/r6_step_timeline.txt
