#!/bin/bash
cd /tmp && export TMPDIR=/tmp
bash $GRAFT_REPO_ROOT/scripts/r6_pgemm4_pmc.sh > $GRAFT_REPO_ROOT/gpurun_out/r6g_pgemm4_pmc.txt 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/r6g_pgemm4_pmc.txt
