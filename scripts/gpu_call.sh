#!/bin/bash
# one small call through the MiniLM-L6 encoder: kernel time against the span of a call (how launch-bound is it?)
# usage: small_pass_trace.sh [B S]   (default 1 16)
B=${1:-1}; S=${2:-16}
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
cat > /tmp/one.py <<PY
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from memex_amd.encoder import Encoder
from memex_amd import weights as W
cfg = W.ALL_DISTILROBERTA_V1
enc = Encoder(cfg, W.synthetic_weights(cfg, 0))
ids = np.random.default_rng(0).integers(1000, cfg.vocab, ($B, $S)).astype(np.int32); lens = np.full(($B,), $S, dtype=np.int32)
for _ in range(20): enc.encode(ids, lens)
PY
rm -rf /tmp/tr; rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tr -- python /tmp/one.py > /dev/null 2>&1
python - > $GRAFT_REPO_ROOT/gpurun_out/r5_small_pass_trace_768_split.txt <<'PY'
import csv, glob
k = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]) for r in csv.DictReader(open(glob.glob("/tmp/tr/**/*_kernel_trace.csv", recursive=True)[0]))))
# last call: the kernels after the last token_map
idx = [i for i, x in enumerate(k) if "token_map" in x[2]]
last = k[idx[-1]:]
busy = sum(e - s for s, e, _ in last) / 1e3
span = (last[-1][1] - last[0][0]) / 1e3
print("kernels in a call: %d, busy %.1f us, first start to last end %.1f us" % (len(last), busy, span))
for (s, e, n), (s2, _, _) in zip(last, last[1:] + [last[-1]]):
    print("%-42s %7.1f us, gap to next %6.1f us" % (n, (e - s) / 1e3, (s2 - e) / 1e3))
calls = [k[a][0] for a in idx]
print("call period (token_map to token_map), us:", [round((b - a) / 1e3) for a, b in zip(calls[-6:], calls[-5:])])
PY
cat $GRAFT_REPO_ROOT/gpurun_out/r5_small_pass_trace_768_split.txt
