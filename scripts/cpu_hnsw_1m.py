"""One-off: the HNSW CPU baseline of bench.py (reference parameters M=16, efc=200, ef=32, one search thread)
at 1M rows on both corpora -> JSON on stdout.  The default bench line runs it at 100k rows only because the
(untimed, parallel) graph build takes minutes at 1M even on 128 cores."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
print(json.dumps(bench.cpu_hnsw(384, 256, 10, rows)))
