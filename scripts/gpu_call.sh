#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_centred_gpu.py -m gpu -x -q -k "random_cones" 2>&1 | tail -6 | tee gpurun_out/r6hh_cones.txt
