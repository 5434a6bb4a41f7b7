#!/bin/bash
# round 4, call 4: the whole GPU suite after the cliff / robustness work (wild-norm rows on the side list, batched EXACT
# path, reversible demotion, RCCL self-test + run-time fallback, checkpoint-like encoder weights, tail2 out of the library)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -rP 2>&1 | grep -v "^$" | tail -60 > gpurun_out/r4_gputests.txt
cat gpurun_out/r4_gputests.txt
