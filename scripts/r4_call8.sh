#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -rf 2>&1 | grep -v "^$" | grep -v "^E    \|^    \|^>" | tail -40 > gpurun_out/r4_gputests3.txt
cat gpurun_out/r4_gputests3.txt
