#!/usr/bin/env python3
"""bench.py -- headline benchmark of the memex MI355X path (contract: see the task brief).

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): a 10M x 384-d f32
synthetic corpus resident in HBM, query batch 256, top-10, exact cosine search through the C ABI
(`mx_index_search_device`).  One "step" = one 256-query batch answered end to end (query prep,
streaming scan, candidate select, f64 rescoring, ordering; plus the RCCL all-gather + merge when
N > 1).  Inputs are resident in HBM before the timed region.

N > 1 (`python -m torch.distributed.run ... bench.py --gpus N`): STRONG scaling on the same 10M-row
corpus -- rank r owns rows [r*10M/N, (r+1)*10M/N) with global ids, every rank answers the same
256 queries on its shard, one all-gather of the per-shard top-10 (ids + dists) and a merge kernel
give every rank the global answer.

Printed JSON (one line, rank 0): metric/value/unit per the contract + `roofline` for the scan
kernel (algorithmic bytes / HIP-event time of the kernel on the library's stream) + `cpu_baseline`
(the C oracle on the host cores, bounded sample, rank 0 at N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rows", type=int, default=10_000_000, help="corpus rows (whole job)")
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--scan", choices=["bf16", "f32"], default="bf16",
                    help="what the scan kernel streams: the bf16 filter copy (default) or the f32 rows")
    ap.add_argument("--alt-steps", type=int, default=20,
                    help="N=1 only: extra untimed-for-the-metric steps on the OTHER scan kernel, reported beside the main one (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    ap.add_argument("--recall-queries", type=int, default=4, help="queries re-answered on the EXACT path")
    ap.add_argument("--ingest-chunks", type=int, default=4096, help="512-token chunks per GPU for the ingest leg (0 = skip)")
    return ap.parse_args()


def cpu_baseline(dim: int, batch: int, k: int, rows_total: int, target_s: float):
    """Time the C oracle (oracle/cosine_oracle.c: DistCosine brute force, OpenMP over queries) on a
    bounded sample of the same workload and scale to the full corpus."""
    from oracle.search_oracle import COracle

    orc = COracle()
    cores = orc.num_threads()
    rng = np.random.default_rng(99)
    q = rng.standard_normal((batch, dim), dtype=np.float32)
    # calibrate on a small slab, then size the sample for ~target_s
    cal_rows = 2000
    x = rng.standard_normal((cal_rows, dim), dtype=np.float32)
    t0 = time.perf_counter()
    orc.search(x, q, k)
    dt = max(time.perf_counter() - t0, 1e-4)
    rows = int(min(max(cal_rows * target_s / dt, cal_rows), 1_600_000))
    x = rng.standard_normal((rows, dim), dtype=np.float32)
    t0 = time.perf_counter()
    orc.search(x, q, k)
    dt = time.perf_counter() - t0
    pairs_per_s = batch * rows / dt
    return {
        "value": pairs_per_s / rows_total,  # queries/s against the full corpus at this pair rate
        "unit": "queries/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{batch} queries x {rows} rows x {dim}-d in {dt:.2f}s, scaled linearly to {rows_total} rows",
    }


MFMA_PEAK_TFLOPS = 2500.0  # dense bf16, MI355X_MICROARCH.md


def encoder_cpu_baseline(cfg, chunks: int = 16):
    """libtorch CPU f32 forward of the same architecture (transformers.BertModel eager, all host
    threads): the closest available proxy for rust-bert's `model.encode` (embedding.rs:109), which
    drives the same operator library through tch.  Bounded sample."""
    try:
        import torch
        from transformers import BertConfig, BertModel
        hc = BertConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers,
                        num_attention_heads=cfg.heads, intermediate_size=cfg.ffn, max_position_embeddings=cfg.max_pos)
        m = BertModel(hc, add_pooling_layer=False).eval()
        ids = torch.randint(1000, cfg.vocab, (chunks, 512))
        with torch.no_grad():
            m(input_ids=ids[:2])
            # bounded sample of ~10 s: batches of `chunks` until the budget is used
            done, t0 = 0, time.perf_counter()
            while True:
                m(input_ids=ids)
                done += chunks
                dt = time.perf_counter() - t0
                if dt >= 10.0 or done >= 64 * chunks:
                    break
        return {"value": done / dt, "unit": "chunks/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": f"{done} x 512-token chunks (batches of {chunks}), transformers.BertModel f32 eager on CPU, {dt:.2f}s"}
    except Exception as e:  # the baseline is optional; never fail the bench for it
        return {"error": repr(e)}


def ingest_leg(chunks: int, dev: int, world: int, cpu_too: bool = True):
    """BASELINE.json configs[4] shape: 512-token chunks, all-MiniLM-L6-v2 architecture with seeded
    synthetic weights (no checkpoints offline), bf16 MFMA encoder, data-parallel replicas (no
    collective).  Reported next to the headline metric; not part of `value`."""
    import torch
    import torch.distributed as dist
    from memex_amd import weights as W
    from memex_amd.encoder import Encoder

    cfg = W.ALL_MINILM_L6_V2
    enc = Encoder(cfg, W.synthetic_weights(cfg, 0), device=dev)
    g = torch.Generator(device="cuda")
    g.manual_seed(77)
    ids = torch.randint(1000, cfg.vocab, (chunks, 512), device="cuda", dtype=torch.int32, generator=g)
    lens = torch.full((chunks,), 512, device="cuda", dtype=torch.int32)
    out = torch.zeros((chunks, cfg.hidden), device="cuda")
    torch.cuda.synchronize()
    enc.encode_device(ids[:256], lens[:256], out[:256])  # warm-up
    enc.reset_stats()
    enc.set_profiling(True)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    reps = 2
    for _ in range(reps):
        enc.encode_device(ids, lens, out)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    st = enc.stats()
    # ragged variant of BASELINE configs[4] (lengths U[64, 512]); one pass, reported only
    ragged = None
    if world == 1:
        rl = torch.randint(64, 513, (chunks,), device="cuda", dtype=torch.int32, generator=g)
        enc.reset_stats()
        torch.cuda.synchronize()
        tr = time.perf_counter()
        enc.encode_device(ids, rl, out)
        torch.cuda.synchronize()
        dr = time.perf_counter() - tr
        sr = enc.stats()
        ragged = {"value": chunks / dr, "unit": "chunks/s", "tokens_per_s": sr.tokens / dr,
                  "tflops": sr.flops / (sr.gpu_ms / 1e3) / 1e12 if sr.gpu_ms > 0 else 0.0,
                  "lengths": "uniform in [64, 512]"}
    enc.close()
    tf = st.flops / (st.gpu_ms / 1e3) / 1e12 if st.gpu_ms > 0 else 0.0
    cpu = None
    if world == 1 and cpu_too:
        cpu = encoder_cpu_baseline(cfg)
    return {
        "cpu_baseline": cpu,
        "metric": "ingest chunks/sec (512-token chunks, all-MiniLM-L6-v2 shape, bf16 MFMA)",
        "value": chunks * reps * world / dt,
        "unit": "chunks/s",
        "chunks_per_gpu": chunks * reps,
        "gflop_per_chunk": st.flops / max(1, st.sequences) / 1e9,
        "ragged": ragged,
        "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": tf / MFMA_PEAK_TFLOPS, "note": "rank-0 GPU time by HIP events on the encoder stream"},
    }


def traffic_from_profile(rows_total: int, dim: int, world: int, scan: str):
    """HBM bytes per launch of the scan kernel from the committed PMC pass (profiles/, FETCH_SIZE x2
    gfx950 correction + WRITE_SIZE, separate --pmc runs).  bench.py cannot collect PMCs itself, so
    this is only reported when the committed profile matches the workload being run."""
    path = os.path.join(ROOT, "profiles", "r1_scan16_traffic.json" if scan == "bf16" else "r1_scan_traffic.json")
    if world != 1 or rows_total != 10_000_000 or dim != 384 or not os.path.exists(path):
        return None
    try:
        with open(path) as f:
            return float(json.load(f)["traffic_bytes_per_launch"])
    except Exception:
        return None


def main():
    a = parse()
    import torch
    import torch.distributed as dist

    from memex_amd import _lib
    from memex_amd.index import FlatIndex, merge_topk_packed_device, packed_result_block

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.gpus > 1 and world == 1:
        raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    # MEMEX_BENCH_ONE_DEVICE=1 is a wiring check for boxes with a single GPU: all ranks share device 0
    # and the collectives go through gloo (RCCL refuses two ranks on one device).  Never a benchmark.
    one_device = os.environ.get("MEMEX_BENCH_ONE_DEVICE") == "1"
    dev = local_rank if (world > 1 and not one_device) else 0
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))

    # ---- corpus shard in HBM (generated on device in blocks; ids are global)
    rows_total = a.rows
    lo = rows_total * rank // world
    hi = rows_total * (rank + 1) // world
    n_local = hi - lo
    idx = FlatIndex(a.dim, key=None, device=dev)
    if a.scan == "f32":
        idx.set_filter_copy(False)
    idx.reserve(n_local)
    idx.set_id_offset(lo)
    gen = torch.Generator(device="cuda")
    block = 1_000_000
    # the corpus is defined by GLOBAL 1M-row blocks (seed = 1234 + block index): every N sees the
    # same 10M rows, a rank generates the blocks that overlap its range and keeps its slice
    for gb in range(lo // block, (hi + block - 1) // block):
        g0 = gb * block
        nb = min(block, rows_total - g0)
        gen.manual_seed(1234 + gb)
        xb = torch.randn((nb, a.dim), device="cuda", dtype=torch.float32, generator=gen)
        s0, s1 = max(lo, g0) - g0, min(hi, g0 + nb) - g0
        part = xb[s0:s1].contiguous()
        torch.cuda.synchronize()
        idx.add_device(part)
        del xb, part
    torch.cuda.empty_cache()
    gq = torch.Generator(device="cuda")
    gq.manual_seed(4321)
    q = torch.randn((a.batch, a.dim), device="cuda", dtype=torch.float32, generator=gq)
    k = a.k
    # ids and dists live in one block so that the N>1 exchange is ONE all-gather (B*k*12 bytes per rank)
    block, ids, dists = packed_result_block(a.batch, k, "cuda")
    scores = torch.zeros((a.batch, k), device="cuda", dtype=torch.float32)
    nf = torch.zeros((a.batch,), device="cuda", dtype=torch.int32)
    if world > 1:
        g_block = torch.zeros((world, block.numel()), device="cuda", dtype=torch.uint8)
        m_ids = torch.zeros((a.batch, k), device="cuda", dtype=torch.int64)
        m_dists = torch.zeros((a.batch, k), device="cuda", dtype=torch.float32)
        m_scores = torch.zeros((a.batch, k), device="cuda", dtype=torch.float32)
    torch.cuda.synchronize()

    def step():
        idx.search_device(q, k, ids, scores, dists, nf)  # blocks until results are in HBM
        if world > 1:
            if one_device:  # gloo: gather through host memory
                parts = [torch.empty_like(block, device="cpu") for _ in range(world)]
                dist.all_gather(parts, block.cpu())
                g_block.copy_(torch.stack(parts).to(g_block.device))
            else:
                dist.all_gather_into_tensor(g_block, block)
            torch.cuda.synchronize()
            merge_topk_packed_device(dev, g_block, world, a.batch, k, m_ids, m_dists, m_scores)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(warmup, steps):
        for _ in range(warmup):
            step()
        idx.reset_stats()
        idx.set_profiling(True)
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        dt_ = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt_], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_ = float(t.item())
        st_ = idx.stats()
        idx.set_profiling(False)
        return dt_, st_

    def roofline(st_, scan):
        scan_s = st_.scan_ms / 1e3
        achieved = (st_.scan_bytes / scan_s / 1e9) if scan_s > 0 else 0.0
        launches = max(1, st_.scan_launches)
        elem = 2 if scan == "bf16" else 4
        tflops = (2.0 * a.batch * (st_.scan_bytes / elem) / scan_s / 1e12) if scan_s > 0 else 0.0
        kc = (a.dim + 127) // 128
        return {
            "bound": "hbm",
            "kernel": f"mx::scan16_kernel<{kc},1> (bf16 filter copy, main stage)" if scan == "bf16"
                      else f"mx::scan_kernel<{kc},1> (f32 rows, main stage)",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "bytes_per_launch": st_.scan_bytes / launches,
            "ms_per_launch": st_.scan_ms / launches,
            "traffic": traffic_from_profile(rows_total, a.dim, world, scan),
            "mfma_tflops": tflops,
            "mfma_frac": tflops / MFMA_PEAK_TFLOPS,
        }

    dt, st = timed(a.warmup, a.steps)
    ids_main = ids.clone()
    alt = None
    if world == 1 and a.alt_steps > 0:
        # the same job on the other scan kernel (results must be identical: same filter arithmetic)
        other = "f32" if a.scan == "bf16" else "bf16"
        idx.set_filter_copy(other == "bf16")
        dt2, st2 = timed(3, a.alt_steps)
        alt = {"scan": other, "value": a.batch * a.alt_steps / dt2, "unit": "queries/s", "steps": a.alt_steps,
               "ms_per_step": dt2 / a.alt_steps * 1e3, "ids_equal_main_run": bool(torch.equal(ids, ids_main)),
               "roofline": roofline(st2, other)}
        idx.set_filter_copy(a.scan == "bf16")

    # ---- recall@10 against the all-f64 EXACT path on a few queries (oracle-level parity at
    # smaller sizes lives in tests/; the EXACT path is itself oracle-checked there)
    recall = None
    if rank == 0 and a.recall_queries > 0:
        nq = min(a.recall_queries, a.batch)
        final_ids = (m_ids if world > 1 else ids)[:nq].clone()
        if world == 1:
            e_ids = torch.zeros((nq, k), device="cuda", dtype=torch.int64)
            e_sc = torch.zeros((nq, k), device="cuda", dtype=torch.float32)
            e_di = torch.zeros((nq, k), device="cuda", dtype=torch.float32)
            e_nf = torch.zeros((nq,), device="cuda", dtype=torch.int32)
            idx.set_search_mode(_lib.MX_SEARCH_EXACT)
            idx.search_device(q[:nq].contiguous(), k, e_ids, e_sc, e_di, e_nf)
            idx.set_search_mode(_lib.MX_SEARCH_AUTO)
            hit = 0
            for b in range(nq):
                hit += len(set(final_ids[b].tolist()) & set(e_ids[b].tolist()))
            recall = hit / float(nq * k)
            recall_exact_order = bool(torch.equal(final_ids, e_ids))
        else:
            recall_exact_order = None
            # N > 1: no rank holds the whole corpus; check the merge's invariants instead (lists ordered by
            # (dist, id), ids unique and inside the corpus, every list full)
            md, mi = m_dists.cpu(), m_ids.cpu()
            merged_ok = bool((md[:, 1:] >= md[:, :-1]).all()) and bool(((mi >= 1) & (mi <= rows_total)).all()) and \
                all(len(set(r.tolist())) == k for r in mi)
    if world > 1:
        dist.barrier()
    ingest = ingest_leg(a.ingest_chunks, dev, world, not a.no_cpu_baseline) if a.ingest_chunks > 0 else None

    if rank == 0:
        out = {
            "metric": "queries/sec, exact cosine top-10 (recall@10 = 1.0) on 10M x 384-d f32",
            "value": a.batch * a.steps / dt,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32 corpus, bf16 MFMA filter + f64 rescoring",
            "scan": a.scan,
            "filter_copy_bytes": int(st.filter_copy_bytes),
            "data": "synthetic",
            "config": {"workload": f"{rows_total}x{a.dim} f32 corpus in HBM, query batch {a.batch}, top-{k}",
                       "rows_per_gpu": n_local, "parallelism": f"row-shard x{world}" if world > 1 else "single GPU"},
            "recall_at_10": recall,
            "ids_equal_exact_path": recall_exact_order if rank == 0 and a.recall_queries > 0 else None,
            "merged_lists_ok": merged_ok if world > 1 and a.recall_queries > 0 else None,
            "fallback_queries": int(st.fallback_queries),
            "candidates_per_query": st.candidates / max(1, st.queries),
            "roofline": roofline(st, a.scan),
        }
        if alt is not None:
            out["other_scan"] = alt
        if ingest is not None:
            out["ingest"] = ingest
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.dim, a.batch, k, rows_total, a.cpu_seconds)
        print(json.dumps(out), flush=True)
    idx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
