#!/bin/bash
# round 6: the profile passes behind profiles/r6_* at HEAD (centred copy on accumulator initial values; scan8_kernel<KC, MODE, QG, CEN>)
cd $GRAFT_REPO_ROOT
timeout 3300 bash scripts/profile_search.sh r6q > gpurun_out/r6q_profile.log 2>&1; echo "profile rc=$?"
ls gpurun_out/profiles_r6q | head -40
