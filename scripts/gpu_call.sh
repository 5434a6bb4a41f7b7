#!/bin/bash
# round 6: A/B on one box -- centred int8 collect pass, a_c reads behind the quick test (new) against all in one batch (old)
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
cp memex_amd/libmemex_hip.so /tmp/new.so
{
for rep in 1 2 3; do
  for v in new old; do
    if [ $v = old ]; then cp build_ub/libmemex_hip_old.so memex_amd/libmemex_hip.so; else cp /tmp/new.so memex_amd/libmemex_hip.so; fi
    echo -n "$v: "; timeout 600 python scripts/gpu_enc_like.py 10000000 40 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']), r['ms_per_step'], r['ms_outside_collect_launch'], r['candidates_per_query'])"
  done
done
cp /tmp/new.so memex_amd/libmemex_hip.so
} 2>&1 | tee gpurun_out/r6j_ab.txt
