"""The enc_like leg of bench.py alone (rows expanded from encoder outputs: a narrow cone, the hard end for the filter
certificates), for kernel traces: python scripts/gpu_enc_like.py [rows] [steps]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MEMEX_HIP_SPIN", "1")
import bench
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bench.MIN_SECONDS = 0.0
out = bench.enc_like_leg(rows, 100_000, 256, 10, steps)
out.pop("roofline", None)
print(json.dumps(out))
