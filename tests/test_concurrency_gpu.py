"""Concurrent use of one resident index the way memex uses it: API threads search (one query each: handlers.rs:61-81, combined
in the library), worker threads append (tasks.rs:59), and in between the copy kind changes, the index is saved, capacity is
reserved.  Rows are append-only, an append is atomic, so every answer must be THE exact top-k of some prefix of the rows that
ends at an append boundary between "appends finished before the search started" and "appends started before it returned" --
checked against the oracle after the threads are done (a linearisability check, not a smoke test).

MEMEX_TEST_SOAK=n runs n times as long."""
import os
import threading
import time

import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("d,cone,shards", [(64, False, 1), (384, True, 1), (96, False, 3)])
def test_searches_are_linearisable_under_appends(d, cone, shards, oracle, lib_built, tmp_path):
    from memex_amd.index import FlatIndex
    soak = int(os.environ.get("MEMEX_TEST_SOAK", "0"))
    seconds = 3.0 * (1 + soak)
    rng0 = np.random.default_rng(d)
    axis = rng0.standard_normal(d).astype(np.float32)
    axis /= np.linalg.norm(axis)

    def make_rows(rng, n):
        X = rng.standard_normal((n, d))
        if cone:
            X = axis + X * (0.8 / np.sqrt(d))
        return (X * rng.uniform(0.5, 2.0, (n, 1))).astype(np.float32)

    # shards > 1: the in-library sharded index (rows dealt to the shards in blocks; here all on device 0) -- an append spans
    # several shards and must still be atomic for a concurrent search
    idx = FlatIndex(d) if shards == 1 else FlatIndex(d, devices=[0] * shards, block_rows=512)
    first = make_rows(rng0, 3000)
    idx.add(first)
    batches = [first]                    # appended batches, in id order
    bounds = [3000]                      # row count after each finished append
    started = [3000]                     # row count once the append in flight is done (>= bounds[-1])
    book = threading.Lock()              # the test's own bookkeeping; the appends themselves are serialised by it too
    stop = threading.Event()
    records, errors = [], []

    def adder(seed):
        rng = np.random.default_rng(seed)
        while not stop.is_set():
            X = make_rows(rng, int(rng.choice([1, 7, 64, 65, 500, 2000])))
            with book:
                started[0] = bounds[-1] + len(X)
                try:
                    assert idx.add(X) == bounds[-1] + 1
                except Exception as e:  # noqa: BLE001
                    errors.append(("add", repr(e)))
                    return
                batches.append(X)
                bounds.append(bounds[-1] + len(X))
            time.sleep(float(rng.uniform(0.004, 0.012)))

    def searcher(seed):
        rng = np.random.default_rng(seed)
        while not stop.is_set():
            B = int(rng.choice([1, 1, 1, 3, 40]))
            k = int(rng.choice([1, 5, 10]))
            Q = rng.standard_normal((B, d)).astype(np.float32)
            if cone:
                Q = (axis + Q * (0.8 / np.sqrt(d))).astype(np.float32)
            lo = bounds[-1]
            try:
                ids, sc, di, nf = idx.search(Q, k)
            except Exception as e:  # noqa: BLE001
                errors.append(("search", repr(e)))
                return
            hi = started[0]
            records.append((Q, k, ids.copy(), di.copy(), nf.copy(), lo, hi))

    def meddler(seed):
        rng = np.random.default_rng(seed)
        kinds = ["i8", "bf16", True, "bf16"]
        while not stop.is_set():
            try:
                op = int(rng.integers(0, 3))
                if op == 0:
                    idx.set_filter_copy(kinds[int(rng.integers(0, len(kinds)))])
                elif op == 1:
                    idx.save(str(tmp_path / "store"))
                else:
                    idx.reserve(bounds[-1] + int(rng.integers(1, 20000)))
            except Exception as e:  # noqa: BLE001
                errors.append(("meddle", repr(e)))
                return
            time.sleep(float(rng.uniform(0.01, 0.05)))

    threads = [threading.Thread(target=adder, args=(1,))] + [threading.Thread(target=searcher, args=(10 + i,)) for i in range(6)] + \
        [threading.Thread(target=meddler, args=(99,))]
    for t in threads:
        t.start()
    time.sleep(seconds)
    stop.set()
    for t in threads:
        t.join(timeout=60)
    assert not any(t.is_alive() for t in threads), "a thread is stuck"
    assert not errors, errors[:3]
    rows = np.concatenate(batches)
    assert len(idx) == len(rows) == bounds[-1]
    # every recorded answer is the exact top-k of an append-boundary prefix inside its window
    checked = 0
    bset = np.asarray(bounds)
    for Q, k, ids, di, nf, lo, hi in records[:: max(1, len(records) // (120 * (1 + soak)))]:
        cands = bset[(bset >= lo) & (bset <= hi)]
        assert len(cands) >= 1, (lo, hi)
        ok = False
        for n in cands:
            oi, od, _, onf = oracle.search(rows[:n], Q, k)
            if np.array_equal(ids, oi) and np.array_equal(bits(di), bits(od)) and np.array_equal(nf, onf):
                ok = True
                break
        assert ok, f"answer matches no prefix in [{lo}, {hi}] (candidates {cands.tolist()[:8]})"
        checked += 1
    assert checked >= 50 and len(bounds) > 20, (checked, len(bounds))
    # and the index that went through all that still answers for ALL its rows, on every kind of copy
    Q = make_rows(np.random.default_rng(7), 8)
    oi, od, _, _ = oracle.search(rows, Q, 10)
    for kind in ("i8", "bf16", False):
        idx.set_filter_copy(kind)
        ids, _, di, _ = idx.search(Q, 10)
        np.testing.assert_array_equal(ids, oi)
        np.testing.assert_array_equal(bits(di), bits(od))
    idx.close()
