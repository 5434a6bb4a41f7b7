#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
bash scripts/r6_attention_x3_pmc.sh 2>&1 | tee gpurun_out/r6y_x3_pmc.txt
