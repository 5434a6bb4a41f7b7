#!/bin/bash
# round 6: wiring check of bench.py's N > 1 forms on one GPU (MEMEX_BENCH_ONE_DEVICE=1): one process per rank (torch.distributed.run, gloo on
# one device) and the in-library sharded index (one process).  The numbers mean nothing; the JSON line must come out well-formed.
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export MEMEX_BENCH_ONE_DEVICE=1
for N in 2 4; do
  echo "== torch.distributed.run N=$N"
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2951$N bench.py --gpus $N --steps 5 --warmup 2 --rows 4000000 --ingest-chunks 4096 --no-fallback > gpurun_out/r6r_wiring_pp$N.json 2> gpurun_out/r6r_wiring_pp$N.err; echo "rc=$?"
  tail -1 gpurun_out/r6r_wiring_pp$N.json | python -c "import json,sys; r=json.loads(sys.stdin.read()); print({k: r[k] for k in ('value','n_gpus','steps','ms_per_step','scaling')}, r['config'], list(r['roofline'].keys())[:6], 'ingest' in r)"
done
echo "== in-library N=2"
timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --rows 4000000 --ingest-chunks 4096 --no-fallback > gpurun_out/r6r_wiring_lib2.json 2> gpurun_out/r6r_wiring_lib2.err; echo "rc=$?"
tail -1 gpurun_out/r6r_wiring_lib2.json | python -c "import json,sys; r=json.loads(sys.stdin.read()); print({k: r[k] for k in ('value','n_gpus','steps','ms_per_step','scaling')}, r['config'])"
