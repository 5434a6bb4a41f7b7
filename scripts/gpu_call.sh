#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r5_multirank_wiring.txt
: > $O
export MEMEX_BENCH_ONE_DEVICE=1
COMMON="--rows 2000000 --steps 5 --warmup 2 --alt-steps 0 --side-steps 2 --ingest-chunks 2048 --bge-chunks 0 --cfg2-segments 0 --enc-like-rows 0 --shard-legs 0 --text-docs 0 --no-cpu-baseline --min-seconds 0"
echo "== in-library, 2 logical shards on one device" >> $O
timeout 300 python bench.py --gpus 2 $COMMON 2>&1 | tail -2 | cut -c1-900 >> $O
echo "== per-process, torchrun 2 ranks on one device (gloo exchange)" >> $O
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 $COMMON 2>&1 | tail -2 | cut -c1-900 >> $O
echo "== per-process, 4 ranks" >> $O
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 4 $COMMON 2>&1 | tail -2 | cut -c1-900 >> $O
cat $O
