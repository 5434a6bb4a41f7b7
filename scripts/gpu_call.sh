#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
bash scripts/r6_centred_traffic.sh 2>&1 | tee gpurun_out/r6_scan8_centred_traffic.json
