"""Generates tests/golden/search_golden.npz with the C oracle (oracle/cosine_oracle.c).

The reference (Rust) cannot run here, so these vectors pin the ORACLE + HIP path against each other
and against regressions; the reference-derived pins are the KAT in tests/test_oracle.py.
Inputs are regenerated from seeds (numpy PCG64, same image on both boxes); only outputs are stored.
    python tests/golden/make_search_golden.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.search_oracle import COracle  # noqa: E402

CASES = {  # name: (n, d, B, k, seed)
    "n1000_d3": (1000, 3, 16, 10, 11),
    "n1000_d384": (1000, 384, 16, 10, 12),
    "n1000_d768": (1000, 768, 16, 10, 13),
    "n20000_d384": (20000, 384, 16, 10, 14),
    "n5000_d100_k7": (5000, 100, 8, 7, 15),
}

if __name__ == "__main__":
    orc = COracle()
    out = {}
    for name, (n, d, B, k, seed) in CASES.items():
        rng = np.random.default_rng(seed)
        X = rng.standard_normal((n, d), dtype=np.float32)
        Q = rng.standard_normal((B, d), dtype=np.float32)
        ids, dists, scores, _ = orc.search(X, Q, k)
        out[name + "_cfg"] = np.array([n, d, B, k, seed], dtype=np.int64)
        out[name + "_ids"] = ids
        out[name + "_dists"] = dists
        out[name + "_scores"] = scores
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "search_golden.npz"), **out)
    print("wrote", len(CASES), "cases")
