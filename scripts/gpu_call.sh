#!/bin/bash
# one document at a time (72 windows of <= 128 tokens: a small pass): the encoder step with / without attention_short_kernel
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for v in 0 1 0 1; do echo "== MEMEX_HIP_ATTN_SHORT=$v"; if [ $v = 0 ]; then export MEMEX_HIP_ATTN_SHORT=0; else unset MEMEX_HIP_ATTN_SHORT; fi; timeout 300 python scripts/gpu_text_ingest_profile.py 100 2>&1 | grep -iE "encoder|total|per doc"; done > gpurun_out/r5x_doc_pass_ab.txt
cat gpurun_out/r5x_doc_pass_ab.txt
