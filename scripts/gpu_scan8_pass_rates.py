"""How often scan8_kernel's centred collect pass gets past its quick test (an instrumented build of scan8.hip, build_ub/ only:
counters per wave and half tile -- tested, past the quick test, past the per-row test, groups of 4 rows a per-group bound would pass).
usage: gpu_scan8_pass_rates.py [rows] [steps]   (the library in memex_amd/ must be the instrumented one: scripts/scan8_pass_rates.patch,
which applies to memex_amd/csrc/scan8.hip of commit c819cbd -- the per-row epilogue it counts was replaced by accumulator initial values
in the commit after; profiles/r6_centred_int8_accumulator_init.txt holds what it measured)"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MEMEX_HIP_SPIN", "1")
import bench
from memex_amd import _lib
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
bench.MIN_SECONDS = 0.0
out = bench.enc_like_leg(rows, 100_000, 256, 10, steps)
L = ctypes.CDLL(_lib.LIB_PATH)
c = (ctypes.c_ulonglong * 4)()
assert L.mx_debug_scan8_counts(c, 0) == 0
t, q, f, g = [int(x) for x in c]
print(json.dumps({"qps": round(out["value"]), "scan": out["scan"], "centred": out.get("filter_centred"), "candidates_per_query": out["candidates_per_query"],
                  "wave_halves_tested": t, "past_quick_test": q, "past_row_test": f, "groups_of_4_past_group_test": g,
                  "quick_pass_rate": round(q / max(t, 1), 4), "row_pass_rate": round(f / max(t, 1), 4),
                  "group_pass_rate_among_passing_halves": round(g / max(4 * q, 1), 4)}))
