#!/usr/bin/env python3
"""bench.py -- headline benchmark of the memex MI355X path (contract: see the task brief).

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): a 10M x 384-d f32
synthetic corpus resident in HBM, query batch 256, top-10, exact cosine search through the C ABI
(`mx_index_search_device`).  One "step" = one 256-query batch answered end to end (query prep,
sample scan, threshold, collect scan, candidate select + f32/f64 rescoring + ordering; plus the RCCL
all-gather + merge when N > 1).  Inputs are resident in HBM before the timed region.

N > 1: STRONG scaling on the same 10M-row corpus (north_star: "on a 10M x 384-d corpus ... queries/sec at
1/2/4/8 GPUs"), in either of two forms with identical arithmetic:
  * `python bench.py --gpus N` (one plain process, WORLD_SIZE unset): the form memex itself would run -- its
    api and worker are tasks of ONE process (bin/memex/src/main.rs) -- through the in-library sharded index
    (`mx_index_open_sharded`): rows dealt to the N GPUs in 64k-row blocks, one helper thread + stream per
    shard, ONE RCCL all-gather of the per-shard top-10 blocks, merge on device 0;
  * `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` (one process per GPU): rank r owns
    rows [r*10M/N, (r+1)*10M/N) with global ids, every rank answers the same 256 queries on its shard, ONE
    `all_gather_into_tensor` of the packed blocks and the merge kernel give every rank the global answer.

Printed JSON (ONE line on stdout, rank 0, printed last): metric/value/unit per the contract + `roofline` +
`cpu_baseline`.  EVERY roofline claim of DESIGN.md sits inside `roofline` so that the driver's record alone lets a
reader recompute it: the headline collect-scan kernel (bytes per launch / HIP-event time of the kernel on the
library's stream / 8 TB/s), `section_8d_kernel` (the scan over the f32 rows themselves: SURVEY 8(d)'s N*D*4 bytes
literally), `encoder_minilm` / `encoder_bge` (BASELINE configs[4] and the configs[3] model: 512-token chunks,
GFLOP per chunk x chunks/s / 2.5 PFLOP/s), `encoder_minilm_128tok` (all-MiniLM-L12-v2, the reference's default
model, at its own 128-token window) and `encoder_split_modes` (MX_PREC_BF16X3 / MX_PREC_MIXED: the modes that hold
north_star's 1e-3 on scores, with the bf16 mode's score error beside them).  `cpu_baseline` (rank 0 at N = 1 only): the C oracle's exact brute force on all host cores, the
reference's real algorithm -- HNSW with memex's parameters, one search thread, its recall@10 -- at 100k rows and
(time-bounded) at up to 1M rows, and the encoder's CPU proxies.  The side legs (clustered / anisotropic corpora,
host API, concurrent callers, small batches, query latency, configs[3]'s shard, the 1/8 shards, enc_like_10M,
cfg2, text ingest) go to a SECOND file (`--sides-out`, default gpurun_out/bench_sides.json or ./bench_sides.json)
and to stderr; the stdout line carries a compact `sides` summary of them.
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

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16, same guide
MFMA_PEAK_I8_TOPS = 5000.0  # dense int8 (2x the bf16 rate, same guide)
BLOCK = 1_000_000          # the corpus is defined by GLOBAL 1M-row blocks: every N sees the same rows


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rows", type=int, default=10_000_000, help="corpus rows (whole job)")
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--data", choices=["gaussian", "clustered", "anisotropic"], default="gaussian",
                    help="rows of the headline corpus: i.i.d. N(0,1) (BASELINE's synthetic corpus) or clustered")
    ap.add_argument("--scan", choices=["i8", "bf16", "f32"], default="i8",
                    help="what the scan kernel streams: the int8 filter copy, the bf16 filter copy or the f32 rows")
    ap.add_argument("--alt-steps", type=int, default=20,
                    help="N=1 only: extra steps on the OTHER scan kernel, reported beside the main one (0 = skip)")
    ap.add_argument("--side-steps", type=int, default=20,
                    help="N=1 only: steps of each side leg (clustered data, host API, 10M x 768 shard); 0 = skip them")
    ap.add_argument("--small-steps", type=int, default=None,
                    help="N=1 only: steps of the 1-query and 32-query legs (default: --side-steps; 0 = skip, e.g. under a "
                         "profiler, where their much shorter launches of the same kernel would blur its average)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the brute-force baseline sample")
    ap.add_argument("--hnsw-rows", type=int, default=100_000, help="corpus size of the HNSW CPU baseline (0 = skip)")
    ap.add_argument("--hnsw-big-rows", type=int, default=1_000_000,
                    help="a second, larger HNSW CPU baseline on the clustered corpus (0 = skip); its rows are cut so that the parallel "
                         "graph build, extrapolated from the first run's, fits --hnsw-big-seconds")
    ap.add_argument("--hnsw-big-seconds", type=float, default=100.0)
    ap.add_argument("--short-seqs", type=int, default=131_072, help="N=1 only: 128-token sequences of the all-MiniLM-L12-v2 leg (0 = skip)")
    ap.add_argument("--sides-out", default=None, help="file for the side legs' full reports (default: gpurun_out/bench_sides.json or ./bench_sides.json)")
    ap.add_argument("--recall-queries", type=int, default=4, help="queries re-answered on the EXACT path")
    ap.add_argument("--ingest-chunks", type=int, default=262_144, help="512-token chunks per GPU for the ingest leg (0 = skip)")
    ap.add_argument("--bge-chunks", type=int, default=24_576, help="N=1 only: 512-token chunks of the bge-base-en ingest leg (0 = skip)")
    ap.add_argument("--cfg2-segments", type=int, default=100_000, help="N=1 only: segments of the configs[1] end-to-end leg (0 = skip)")
    ap.add_argument("--precise-chunks", type=int, default=16384, help="N=1 only: 512-token chunks of the MX_PREC_BF16X3 ingest leg (a quarter of them for bge-base; 0 = skip)")
    ap.add_argument("--text-docs", type=int, default=200, help="N=1 only: documents of the text-ingest leg (segmenter + encoder + add; 0 = skip)")
    ap.add_argument("--min-seconds", type=float, default=0.5,
                    help="untimed steps of the same work run in front of every timed region until this much time has passed: the "
                         "package needs ~0.7 s of continuous load to settle at its power-capped clock (profiles/r3_power_scan8.log), "
                         "and 20 steps are 26 ms.  The timed region itself stays EXACTLY --steps steps.  0 = off")
    ap.add_argument("--shard-legs", type=int, default=1, help="N=1 only: the 1.25M-row legs (what each of 8 GPUs runs for configs[2] / [3]); 0 = skip")
    ap.add_argument("--enc-like-rows", type=int, default=10_000_000, help="N=1 only: rows of the leg built from encoder outputs (0 = skip)")
    ap.add_argument("--no-fallback", action="store_true", help="N>1: do not try the other multi-GPU form when this one fails")
    ap.add_argument("--per-process", action="store_true",
                    help="N>1: insist on the one-process-per-GPU form (must be started by torch.distributed.run)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------
# synthetic corpora (generated on the device, block by block, identical for every world size)
# ------------------------------------------------------------------------------------------------
N_CENTRES = 20_000


def clustered_centres(dim: int, dev="cuda"):
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(777)
    c = torch.randn((N_CENTRES, dim), device=dev, generator=g)
    c /= c.norm(dim=1, keepdim=True)
    sig = 0.23 + 0.27 * torch.rand((N_CENTRES, 1), device=dev, generator=g)  # pairwise cosine inside a cluster 0.95 .. 0.8
    return c, sig


def clustered_rows(n: int, dim: int, seed: int, centres, dev="cuda"):
    """n rows = unit centre + sigma * unit-variance noise (intra-cluster cosine 0.8 .. 0.95), 1 % of
    them exact duplicates of another row of the block, random lengths (the path must normalise)."""
    import torch
    c, sig = centres
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    which = torch.randint(0, N_CENTRES, (n,), device=dev, generator=g)
    x = torch.randn((n, dim), device=dev, generator=g) * (sig[which] / dim ** 0.5) + c[which]
    x *= 0.5 + torch.rand((n, 1), device=dev, generator=g)
    ndup = n // 100
    src = torch.randint(0, n, (ndup,), device=dev, generator=g)
    dst = torch.randint(0, n, (ndup,), device=dev, generator=g)
    x[dst] = x[src]
    return x


def anisotropic_rows(n: int, dim: int, seed: int, dev="cuda"):
    """Rows that look more like sentence embeddings than i.i.d. Gaussians do: a decaying spectrum (dimension i scaled by
    (i+1)^-0.5) and a common mean direction (random pairs have cosine ~0.45): most of a unit vector's energy sits in a few
    dimensions -- what an 8-bit quantiser with one step per block dislikes (the rotation)."""
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    spec = torch.arange(1, dim + 1, device=dev, dtype=torch.float32) ** -0.5
    x = torch.randn((n, dim), device=dev, dtype=torch.float32, generator=g) * spec
    x[:, 0] += 0.6 * spec.norm()
    return x


def gaussian_rows(n: int, dim: int, seed: int, dev="cuda"):
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    return torch.randn((n, dim), device=dev, dtype=torch.float32, generator=g)


def fill_index(idx, rows_total: int, dim: int, lo: int, hi: int, data: str):
    """Append global rows [lo, hi) to idx (a rank generates the blocks that overlap its range)."""
    import torch
    centres = clustered_centres(dim) if data == "clustered" else None
    for gb in range(lo // BLOCK, (hi + BLOCK - 1) // BLOCK):
        g0 = gb * BLOCK
        nb = min(BLOCK, rows_total - g0)
        xb = (clustered_rows(nb, dim, 5000 + gb, centres) if data == "clustered" else
              anisotropic_rows(nb, dim, 9000 + gb) if data == "anisotropic" else gaussian_rows(nb, dim, 1234 + gb))
        s0, s1 = max(lo, g0) - g0, min(hi, g0 + nb) - g0
        part = xb[s0:s1].contiguous()
        idx.add_device(part)
        del xb, part
    torch.cuda.empty_cache()


def make_queries(batch: int, dim: int, data: str, seed: int = 4321):
    import torch
    if data == "clustered":  # queries live in clusters too: each has a dense neighbourhood
        return clustered_rows(batch, dim, seed, clustered_centres(dim))
    if data == "anisotropic":
        return anisotropic_rows(batch, dim, seed)
    return gaussian_rows(batch, dim, seed)


# ------------------------------------------------------------------------------------------------
# CPU baselines (rank 0, N = 1): test/bench infrastructure under oracle/, never the thing shipped
# ------------------------------------------------------------------------------------------------
def cpu_bruteforce(dim: int, batch: int, k: int, rows_total: int, target_s: float):
    """The C oracle (oracle/cosine_oracle.c: DistCosine brute force, OpenMP over queries) on a bounded
    sample of the same workload, scaled linearly to the full corpus."""
    from oracle.search_oracle import COracle

    orc = COracle()
    cores = orc.num_threads()
    rng = np.random.default_rng(99)
    q = rng.standard_normal((batch, dim), dtype=np.float32)
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
        "algorithm": "exact brute force, DistCosine arithmetic (oracle/cosine_oracle.c), OpenMP over queries",
        "sample": f"{batch} queries x {rows} rows x {dim}-d in {dt:.2f}s, scaled linearly to {rows_total} rows",
    }


def cpu_hnsw(dim: int, batch: int, k: int, rows: int, corpora=("gaussian", "clustered")):
    """The reference's real search algorithm: HNSW M=16, ef_construction=200, search ef=32, DistCosine,
    one thread per query (local.rs:76,101), restated in oracle/hnsw_baseline.cpp.  QPS at `rows` rows
    (NOT scaled: HNSW cost grows ~log N) and its recall@k against exact search, on both corpora."""
    import torch
    from oracle.hnsw_baseline import HnswBaseline
    from oracle.search_oracle import COracle

    out = []
    for data in corpora:
        if data == "clustered":
            cen = clustered_centres(dim, "cpu")
            x = clustered_rows(rows, dim, 5000, cen, "cpu").numpy()
            q = clustered_rows(batch, dim, 4321, cen, "cpu").numpy()
        else:
            x = gaussian_rows(rows, dim, 1234, "cpu").numpy()
            q = gaussian_rows(batch, dim, 4321, "cpu").numpy()
        t0 = time.perf_counter()
        h = HnswBaseline(x, seed=1, threads=0)
        build_s = time.perf_counter() - t0
        ids, _, sec = h.search(q, k)
        h.close()
        oi = COracle().search(x, q, k)[0]
        rec = float(np.mean([len(set(a) & set(b)) / float(k) for a, b in zip(ids.tolist(), oi.tolist())]))
        out.append({"data": data, "rows": rows, "value": batch / sec, "unit": "queries/s", "cores": 1,
                    "recall_at_10": rec, "build_s": build_s, "build_threads": torch.get_num_threads()})
    return {"kind": "hnsw-port",
            "algorithm": "HNSW M=16 ef_construction=200 ef=32 DistCosine (local.rs:76,101), oracle/hnsw_baseline.cpp; "
                         "search on ONE thread like the reference, parallel build (not timed)",
            "runs": out}


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
            done, t0 = 0, time.perf_counter()
            while True:  # bounded sample of ~10 s
                m(input_ids=ids)
                done += chunks
                dt = time.perf_counter() - t0
                if dt >= 10.0 or done >= 64 * chunks:
                    break
        return {"value": done / dt, "unit": "chunks/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": f"{done} x 512-token chunks (batches of {chunks}), transformers.BertModel f32 eager on CPU, {dt:.2f}s"}
    except Exception as e:  # the baseline is optional; never fail the bench for it
        return {"error": repr(e)}


# ------------------------------------------------------------------------------------------------
# ingest leg (BASELINE.json configs[4])
# ------------------------------------------------------------------------------------------------
def _ingest_calls(enc, ids, lens, chunks: int, call: int):
    done = 0
    while done < chunks:
        n = min(call, chunks - done)
        enc.encode(ids[:n], lens[:n])
        done += n


def ingest_leg(chunks: int, dev: int, world: int, cpu_too: bool = True, model: str = "all-MiniLM-L6-v2", devices=None, seq: int = 512):
    """512-token chunks, the named architecture with seeded synthetic weights (no checkpoints offline), bf16
    MFMA encoder, data-parallel replicas (no collective).  The timed region starts from token ids in HOST
    memory and ends with the f32 embeddings back in host memory (H2D of ids, D2H of outputs included: what
    the worker's embed step sees), in calls of `call` chunks.  `devices` (plain-process N > 1): one replica
    and one host thread per device.  Reported next to the headline metric; not part of `value`."""
    import threading
    import torch
    import torch.distributed as dist
    from memex_amd import weights as W
    from memex_amd.encoder import Encoder

    cfg = {"all-MiniLM-L6-v2": W.ALL_MINILM_L6_V2, "all-MiniLM-L12-v2": W.ALL_MINILM_L12_V2, "bge-base-en": W.BGE_BASE_EN}[model]
    call = (16384 if cfg.hidden == 384 else 4096) * (512 // seq)
    wts = W.pack_weights(W.synthetic_weights(cfg, 0), cfg)
    rng = np.random.default_rng(77)
    ids = rng.integers(1000, cfg.vocab, size=(call, seq), dtype=np.int32)   # one call's worth, reused (content does not matter)
    lens = np.full((call,), seq, dtype=np.int32)
    devs = list(devices) if devices else [dev]
    encs = [Encoder(cfg, wts, device=d) for d in devs]
    for e in encs:
        e.encode(ids[:256], lens[:256])  # warm-up
        e.reset_stats()
        e.set_profiling(True)
    if world > 1:
        dist.barrier()
    if len(encs) == 1:
        t0 = time.perf_counter()
        _ingest_calls(encs[0], ids, lens, chunks, call)
        dt = time.perf_counter() - t0
    else:  # ctypes releases the GIL inside mx_encoder_encode: the replicas run concurrently
        gate = threading.Barrier(len(encs) + 1)
        th = [threading.Thread(target=lambda e=e: (gate.wait(), _ingest_calls(e, ids, lens, chunks, call))) for e in encs]
        for t in th:
            t.start()
        gate.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    enc = encs[0]
    st = enc.stats()
    ragged = None
    if world == 1 and len(encs) == 1 and cfg.hidden == 384 and seq == 512:  # lengths U[64, 512] (BASELINE configs[4]'s ragged variant); reported only
        rl = rng.integers(64, 513, size=(call,), dtype=np.int32)
        enc.reset_stats()
        tr = time.perf_counter()
        for _ in range(4):
            enc.encode(ids, rl)
        dr = time.perf_counter() - tr
        sr = enc.stats()
        rtf = sr.flops / (sr.gpu_ms / 1e3) / 1e12 if sr.gpu_ms > 0 else 0.0
        ragged = {"value": 4 * call / dr, "unit": "chunks/s", "tokens_per_s": sr.tokens / dr,
                  "tflops": rtf, "frac": rtf / MFMA_PEAK_TFLOPS,
                  "roofline": {"bound": "mfma", "achieved": rtf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": rtf / MFMA_PEAK_TFLOPS},
                  "lengths": "uniform in [64, 512]"}
    for e in encs:
        e.close()
    tf = st.flops / (st.gpu_ms / 1e3) / 1e12 if st.gpu_ms > 0 else 0.0
    cpu = encoder_cpu_baseline(cfg, 16 if cfg.hidden == 384 else 8) if (world == 1 and len(encs) == 1 and cpu_too and seq == 512) else None
    replicas = world * len(encs)
    return {
        "cpu_baseline": cpu,
        "metric": f"ingest chunks/sec ({seq}-token chunks, {model} shape, bf16 MFMA; host ids in, host embeddings out)",
        "value": chunks * replicas / dt,
        "unit": "chunks/s",
        "replicas": replicas,
        "chunks_per_gpu": chunks,
        "gflop_per_chunk": st.flops / max(1, st.sequences) / 1e9,
        "gpu_only_chunks_per_s": st.sequences / (st.gpu_ms / 1e3) if st.gpu_ms > 0 else 0.0,
        "ragged": ragged,
        "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": tf / MFMA_PEAK_TFLOPS, "note": "GPU time of replica 0 by HIP events on the encoder stream"},
    }


def precise_ingest_leg(chunks_l6: int, chunks_bge: int):
    """The split-operand modes on the ingest shapes, device ids in -> device embeddings out: MX_PREC_BF16X3 (every GEMM / attention
    product as three bf16 MFMA products, f32 hidden state) and MX_PREC_MIXED (round 6: the same with the MLP's two GEMMs on TWO
    fp16 products) -- the modes whose search scores stay within north_star's 1e-3 under checkpoint-like weights (DESIGN.md
    section 4.2).  `algorithmic_tflops` count one product per product; `mfma_frac` the MFMA products actually issued."""
    import dataclasses
    import torch
    from memex_amd import weights as W
    from memex_amd.encoder import Encoder
    out = {}
    for name, base, chunks in (("all-MiniLM-L6-v2", W.ALL_MINILM_L6_V2, chunks_l6), ("bge-base-en", W.BGE_BASE_EN, chunks_bge)):
        if chunks <= 0:
            continue
        out[name] = {}
        H, F = base.hidden, base.ffn
        for prec in ("bf16x3", "mixed", "mixed1"):
            cfg = dataclasses.replace(base, precision=prec)
            enc = Encoder(cfg, W.pack_weights(W.synthetic_weights(cfg, 0), cfg))
            g = torch.Generator(device="cuda")
            g.manual_seed(3)
            B = 256
            ids = torch.randint(1000, cfg.vocab, (B, 512), device="cuda", dtype=torch.int32, generator=g)
            lens = torch.full((B,), 512, device="cuda", dtype=torch.int32)
            emb = torch.zeros((B, cfg.hidden), device="cuda")
            enc.encode_device(ids, lens, emb)
            enc.reset_stats()
            enc.set_profiling(True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(max(1, chunks // B)):
                enc.encode_device(ids, lens, emb)
            dt = time.perf_counter() - t0
            st = enc.stats()
            enc.close()
            tf = st.flops / (st.gpu_ms / 1e3) / 1e12 if st.gpu_ms > 0 else 0.0
            # MFMA products per algorithmic product: 3 everywhere (bf16x3); mixed: 3 in the projections of the attention block and in
            # QK^T, 2 in P.V (P as one bf16 value) and in the MLP
            gemm, attn = 8.0 * H * H + 4.0 * H * F, 4.0 * 512 * H
            mlp = {"mixed": 2.0, "mixed1": 1.0}.get(prec, 3.0)
            mult = 3.0 if prec == "bf16x3" else (3.0 * (8.0 * H * H + 0.5 * attn) + mlp * 4.0 * H * F + 2.0 * 0.5 * attn) / (gemm + attn)
            out[name][prec] = {"value": st.sequences / dt, "unit": "chunks/s", "chunks": int(st.sequences), "algorithmic_tflops": tf,
                               "mfma_products_per_product": mult, "mfma_frac": mult * tf / MFMA_PEAK_TFLOPS}
            del ids, lens, emb
            torch.cuda.empty_cache()
        # what the modes are for, measured here on the device: the cosines BETWEEN embeddings under checkpoint-like weights (outlier
        # dimensions, logits of +-60), against MX_PREC_BF16X3 -- which tests/test_encoder_gpu.py holds against the f64 oracle
        # (pairwise error <= 2.4e-5 on the four test cases, 6.0e-4 on the harshest seed tried; bound 1e-3.  The mixed modes: DESIGN.md 4.2's correction)
        try:
            small = dataclasses.replace(base, layers=min(base.layers, 12), vocab=3000)
            wck = W.checkpoint_like_weights(small, 52)
            rng = np.random.default_rng(52)
            cid = rng.integers(0, small.vocab, (8, 200)).astype(np.int32)
            cln = rng.integers(100, 201, 8).astype(np.int32)
            embs = {}
            for prec in ("bf16", "mixed", "mixed1", "bf16x3"):
                c2 = dataclasses.replace(small, precision=prec)
                with Encoder(c2, W.pack_weights(wck, c2)) as e2:
                    v = e2.encode(cid, cln).astype(np.float64)
                embs[prec] = v / np.linalg.norm(v, axis=1, keepdims=True)
            ref = embs["bf16x3"] @ embs["bf16x3"].T
            out[name]["score_error_vs_bf16x3"] = {"bf16": float(np.abs(embs["bf16"] @ embs["bf16"].T - ref).max()),
                                                  "mixed": float(np.abs(embs["mixed"] @ embs["mixed"].T - ref).max()),
                                                  "mixed1": float(np.abs(embs["mixed1"] @ embs["mixed1"].T - ref).max())}
            out[name]["score_error_note"] = ("max |cos(e_i, e_j)| difference to the bf16x3 mode, 8 x 200 tokens, checkpoint_like_weights "
                                             "(north_star bar on scores: 1e-3)")
        except Exception as e:  # noqa: BLE001
            out[name]["score_error_vs_bf16x3"] = None
            out[name]["score_error_note"] = repr(e)[:200]
    return out


def cfg2_leg(n_seg: int, batch: int, k: int, steps: int):
    """BASELINE.json configs[1] end to end on one GPU: all-MiniLM-L6-v2 shape (seeded weights), `n_seg`
    synthetic segments with lengths U[16, 256] -> mx_encoder_encode_device -> mx_index_add_device (the
    embeddings never leave HBM: embedding.rs:109 -> tasks.rs:59) -> `batch` encoded queries -> top-k
    (local.rs:71-91).  The 100k x 384 corpus (154 MB) sits in the 256 MiB Infinity Cache: the search rate
    is reported only, not read as an HBM roofline point.  Parity of this flow
    against the oracle: tests/test_cfg2_gpu.py."""
    import torch
    from memex_amd import _lib
    from memex_amd import weights as W
    from memex_amd.encoder import Encoder
    from memex_amd.index import FlatIndex

    cfg = W.ALL_MINILM_L6_V2
    S = 256
    g = torch.Generator(device="cuda")
    g.manual_seed(2024)
    ids = torch.randint(1000, cfg.vocab, (n_seg, S), device="cuda", dtype=torch.int32, generator=g)
    lens = torch.randint(16, S + 1, (n_seg,), device="cuda", dtype=torch.int32, generator=g)
    qids = torch.randint(1000, cfg.vocab, (batch, 32), device="cuda", dtype=torch.int32, generator=g)
    qlens = torch.randint(4, 33, (batch,), device="cuda", dtype=torch.int32, generator=g)
    vec = torch.zeros((n_seg, cfg.hidden), device="cuda")
    q = torch.zeros((batch, cfg.hidden), device="cuda")
    enc = Encoder(cfg, W.pack_weights(W.synthetic_weights(cfg, 0), cfg))
    idx = FlatIndex(cfg.hidden)
    call = 16384
    enc.encode_device(ids[:256], lens[:256], vec[:256])  # warm-up
    enc.reset_stats()
    enc.set_profiling(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b0 in range(0, n_seg, call):
        enc.encode_device(ids[b0:b0 + call], lens[b0:b0 + call], vec[b0:b0 + call])
    t_embed = time.perf_counter() - t0
    st = enc.stats()
    t0 = time.perf_counter()
    idx.add_device(vec)
    t_add = time.perf_counter() - t0
    t0 = time.perf_counter()
    enc.encode_device(qids, qlens, q)
    t_q = time.perf_counter() - t0
    bufs = SearchBuffers(batch, k)

    def step():
        idx.search_device(q, k, bufs.ids, bufs.scores, bufs.dists, bufs.nf)
    step()
    first_st = idx.stats()   # (a demotion of the int8 copy happens on the first batch, before the timed region resets the counters)
    dt, sst = timed_steps(idx, step, torch.cuda.synchronize, 3, steps, 1)
    nq = min(4, batch)
    e = SearchBuffers(nq, k)
    idx.set_search_mode(_lib.MX_SEARCH_EXACT)
    idx.search_device(q[:nq].contiguous(), k, e.ids, e.scores, e.dists, e.nf)
    same = bool(torch.equal(bufs.ids[:nq], e.ids))
    idx.close()
    enc.close()
    tf = st.flops / (st.gpu_ms / 1e3) / 1e12 if st.gpu_ms > 0 else 0.0
    out = {"workload": f"all-MiniLM-L6-v2 shape, {n_seg} synthetic segments (lengths U[16,256]) embedded on the GPU, "
                       f"appended from HBM, {batch} encoded queries, top-{k}",
           "embed_segments_per_s": n_seg / t_embed, "embed_tokens_per_s": st.tokens / t_embed, "embed_tflops": tf,
           "embed_mfma_frac": tf / MFMA_PEAK_TFLOPS, "add_device_ms": t_add * 1e3, "encode_queries_ms": t_q * 1e3,
           "search_value": batch * steps / dt, "search_unit": "queries/s", "search_ms_per_step": dt / steps * 1e3,
           "search_note": "corpus is Infinity-Cache resident (154 MB): report only, not an HBM roofline point",
           "fallback_queries": int(sst.fallback_queries), "retry_queries": int(sst.retry_queries),
           "candidates_per_query": sst.candidates / max(1, sst.queries), "scan": scan_of(sst),
           "filter_demotions": int(first_st.filter_demotions + sst.filter_demotions), "filter_centred": int(sst.filter_centred),
           "approx_err_bound": sst.approx_err_bound, "ids_equal_exact_path": same}
    del ids, lens, vec, q, bufs
    torch.cuda.empty_cache()
    return out


def _synthetic_vocab_and_docs(n_docs: int, chars: int, seed: int = 0):
    """A BERT-sized WordPiece vocabulary (30522 entries: word-like stems and ## continuations over the alphabet) and `n_docs`
    documents of ~`chars` characters (the reference's example document, state_of_the_union_2023.json, holds 42,099): Zipf-distributed
    vocabulary words, out-of-vocabulary words that split into pieces, punctuation, capitals."""
    rng = np.random.default_rng(seed)
    letters = list("etaoinshrdlcumwfgypbvkjxqz")
    word = lambda n: "".join(rng.choice(letters, size=n))  # noqa: E731
    specials = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    stems, conts = set(), set()
    while len(stems) < 20000:
        stems.add(word(int(rng.integers(1, 8))))
    while len(conts) < 10000:
        conts.add("##" + word(int(rng.integers(1, 5))))
    stems = sorted(stems)
    vocab = specials + letters + ["##" + c for c in letters] + stems + sorted(conts) + list("!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~") + \
        [str(i) for i in range(10)]
    seen = set()
    vocab = [v for v in vocab if not (v in seen or seen.add(v))]
    while len(vocab) < 30522:
        vocab.append(f"[unused{len(vocab)}]")
    oov = np.array([word(int(n)) for n in rng.integers(4, 12, 4096)])       # out-of-vocabulary words: split into pieces or [UNK]
    marks = np.array([",", ".", "'s", "--", "(", ")", "2023", "Biden's"])
    stems_a = np.array(stems)
    docs = []
    for _ in range(n_docs):
        m = chars // 4                                                         # more words than needed, cut to size below
        r = rng.random(m)
        ws = np.where(r < 0.7, stems_a[rng.zipf(1.3, m) % len(stems_a)],
                      np.where(r < 0.95, oov[rng.integers(0, len(oov), m)], marks[rng.integers(0, len(marks), m)]))
        cap = rng.random(m) < 0.1
        text = " ".join(w.capitalize() if c else w for w, c in zip(ws.tolist(), cap.tolist()))
        docs.append(text[:text.rfind(" ", 0, chars)])
    return vocab, docs


def text_ingest_leg(n_docs: int, workers: int = 5, cpu_too: bool = True):
    """The reference's ingest flow from TEXT (BASELINE configs[0] scaled up): `workers` threads
    -- the reference runs up to five ingest tasks at a time, worker/lib.rs:36 -- each take documents off a list and call
    process_embeddings (tasks.rs:9-66): segment_text (native WordPiece, windows 256/86, embedding.rs:155-198) -> model.encode of the
    windows (the reference's default model, all-MiniLM-L12-v2 shape, seeded weights, max_seq_length 128) -> add_vectors.  One
    long-lived embedder and one resident collection; concurrent requests are segmented and embedded together.
    Beside it, on the host cores: the segmenter alone, native against the `tokenizers` package -- the Python binding of the very
    crate the reference calls (lib/libmemex/Cargo.toml:31) -- on the same documents and the same call sequence."""
    import tempfile
    import threading
    from memex_amd import tasks
    from memex_amd.embedding import ModelConfig, SentenceEmbedder
    from memex_amd.storage import get_vector_storage
    from memex_amd.tokenizer import WordPieceTokenizer

    vocab, docs = _synthetic_vocab_and_docs(n_docs, 42_000)
    chars = sum(len(d) for d in docs)
    tok = WordPieceTokenizer(vocab, lowercase=True)
    out = {"workload": f"{n_docs} documents x ~42k characters -> segment_text (256/86) -> all-MiniLM-L12-v2 shape (seeded weights, "
                       f"max_seq_length 128) -> add_vectors, {workers} concurrent process_embeddings callers",
           "documents": n_docs, "characters": chars}
    # ---- the segmenter alone (host code; this is the part the reference's crate can be timed against)
    t0 = time.perf_counter()
    one = [tok.windows(d, 256, 86) for d in docs[:16]]
    t_one = (time.perf_counter() - t0) / max(1, sum(len(d) for d in docs[:16]))
    t0 = time.perf_counter()
    segs = tok.windows_batch(docs, 256, 86)
    t_batch = time.perf_counter() - t0
    assert segs[:16] == one
    n_win = sum(len(s_) for s_ in segs)
    out["windows"] = n_win
    out["segmenter"] = {"native_one_thread_MBps": 1e-6 / t_one, "native_batch_MBps": chars / t_batch / 1e6,
                        "native_batch_docs_per_s": n_docs / t_batch, "host_threads": min(os.cpu_count() or 1, 64, n_docs)}
    if cpu_too:
        try:
            import tempfile as _tf
            from tokenizers import BertWordPieceTokenizer
            with _tf.TemporaryDirectory() as td:
                vp = os.path.join(td, "vocab.txt")
                with open(vp, "w", encoding="utf-8") as f:
                    f.write("\n".join(vocab) + "\n")
                hf = BertWordPieceTokenizer(vp, lowercase=True)
            hf.enable_truncation(max_length=256, stride=86)
            sample = docs[:8]
            t0 = time.perf_counter()
            ref = []
            for d in sample:                                                # embedding.rs:173-195
                e = hf.encode(d, add_special_tokens=False)
                w = [hf.decode(e.ids, skip_special_tokens=True).replace(" ' ", "'")]
                w += [hf.decode(o.ids, skip_special_tokens=True) for o in e.overflowing]
                ref.append(w)
            t_ref = time.perf_counter() - t0
            out["segmenter"]["cpu_baseline"] = {
                "value": sum(len(d) for d in sample) / t_ref / 1e6, "unit": "MB/s of text", "cores": 1, "kind": "reference",
                "sample": f"{len(sample)} of the documents through the `tokenizers` package (binding of the crate the reference links), "
                          f"the call sequence of embedding.rs:173-195, {t_ref:.2f}s",
                "windows_equal_native": ref == segs[:len(sample)]}
        except Exception as e:  # noqa: BLE001 -- the baseline must never fail the bench
            out["segmenter"]["cpu_baseline"] = {"error": repr(e)[:200]}
    # ---- the flow
    th, emb = SentenceEmbedder.spawn(ModelConfig(), tokenizer=tok, allow_synthetic=True)
    with tempfile.TemporaryDirectory() as td:
        client = get_vector_storage("hip://" + td, "bench_text")
        tasks.process_embeddings(client, emb, 0, docs[0])                  # warm-up (encoder set-up, first save)
        nxt, lock, errs = [1], threading.Lock(), []

        def worker():
            while True:
                with lock:
                    i = nxt[0]
                    nxt[0] += 1
                if i >= n_docs:
                    return
                try:
                    tasks.process_embeddings(client, emb, i, docs[i])
                except Exception as e:  # noqa: BLE001
                    errs.append(repr(e))
                    return
        t0 = time.perf_counter()
        ts = [threading.Thread(target=worker) for _ in range(workers)]
        for t_ in ts:
            t_.start()
        for t_ in ts:
            t_.join()
        dt = time.perf_counter() - t0
        hits = tasks.search_docs(client, emb, segs[3][0], 3)               # the text of a stored window finds that window
        out.update({"value": (n_docs - 1) / dt, "unit": "documents/s", "windows_per_s": (n_win - len(segs[0])) / dt,
                    "text_MBps": (chars - len(docs[0])) / dt / 1e6, "seconds": dt, "errors": errs[:3],
                    "query_finds_its_window": bool(hits) and hits[0][0] == tasks.segment_uuid(tasks.document_uuid(3), 0)
                    and hits[0][1] > 0.999})
        client.delete_collection()
    emb.shutdown()
    th.join()
    tok.close()
    return out


def enc_like_leg(rows: int, n_seg: int, batch: int, k: int, steps: int):
    """The int8 default on rows that CAME OUT OF the encoder (the only encoder-produced corpus in the repo, cfg2's
    100k, demotes the copy on its first batch): `n_seg` segments embedded with the all-MiniLM-L6-v2 shape (seeded weights), then
    expanded to `rows` rows by small perturbations (each row = an embedding + noise of relative norm 0.1: cosine 0.995 to its
    source), searched with `batch` encoded queries.  Random-weight embeddings sit in a narrow cone -- denser than a trained
    model's -- so this is the hard end for the int8 certificate; the leg reports which copy the library ends on and what the
    automatic policy cost (demotions, retries, fallbacks)."""
    import torch
    from memex_amd import weights as W
    from memex_amd.encoder import Encoder
    from memex_amd.index import FlatIndex

    cfg = W.ALL_MINILM_L6_V2
    S = 128
    g = torch.Generator(device="cuda")
    g.manual_seed(2025)
    ids = torch.randint(1000, cfg.vocab, (n_seg, S), device="cuda", dtype=torch.int32, generator=g)
    lens = torch.randint(16, S + 1, (n_seg,), device="cuda", dtype=torch.int32, generator=g)
    qids = torch.randint(1000, cfg.vocab, (batch, 32), device="cuda", dtype=torch.int32, generator=g)
    qlens = torch.randint(4, 33, (batch,), device="cuda", dtype=torch.int32, generator=g)
    vec = torch.zeros((n_seg, cfg.hidden), device="cuda")
    q = torch.zeros((batch, cfg.hidden), device="cuda")
    enc = Encoder(cfg, W.pack_weights(W.synthetic_weights(cfg, 0), cfg))
    for b0 in range(0, n_seg, 16384):
        enc.encode_device(ids[b0:b0 + 16384], lens[b0:b0 + 16384], vec[b0:b0 + 16384])
    enc.encode_device(qids, qlens, q)
    enc.close()
    mean_cos = float((vec[:2000] @ vec[:2000].T).mean())
    idx = FlatIndex(cfg.hidden)
    idx.reserve(rows)
    for b0 in range(0, rows, BLOCK):
        nb = min(BLOCK, rows - b0)
        src = torch.randint(0, n_seg, (nb,), device="cuda", generator=g)
        xb = vec[src] + (0.1 / cfg.hidden ** 0.5) * torch.randn((nb, cfg.hidden), device="cuda", generator=g)
        idx.add_device(xb)
        del xb, src
    kind0 = scan_of(idx.stats())
    bufs = SearchBuffers(batch, k)

    def step():
        idx.search_device(q, k, bufs.ids, bufs.scores, bufs.dists, bufs.nf)
    step()
    first = idx.stats()
    dt, st = timed_steps(idx, step, torch.cuda.synchronize, 3, steps, 1)
    out = leg_report(st, dt, steps, f"{rows}x{cfg.hidden} rows expanded from {n_seg} encoder outputs (all-MiniLM-L6-v2 shape, seeded "
                                    f"weights; mean pairwise cosine of the embeddings {mean_cos:.2f}), {batch} encoded queries, top-{k}",
                     cfg.hidden, batch, rows)
    out["scan_before_first_batch"] = kind0
    out["filter_demotions"] = int(first.filter_demotions + st.filter_demotions)
    out["first_batch"] = {"retry_queries": int(first.retry_queries), "fallback_queries": int(first.fallback_queries)}
    # the answers on the (centred) filter copy against the library's all-f64 EXACT path, a few queries
    from memex_amd import _lib
    nq = min(4, batch)
    e = SearchBuffers(nq, k)
    idx.set_search_mode(_lib.MX_SEARCH_EXACT)
    idx.search_device(q[:nq].contiguous(), k, e.ids, e.scores, e.dists, e.nf)
    out["ids_equal_exact_path"] = bool(torch.equal(bufs.ids[:nq], e.ids)) and bool(torch.equal(bufs.dists[:nq], e.dists))
    idx.close()
    del idx, vec, q, bufs, ids, e
    torch.cuda.empty_cache()
    return out


def traffic_from_profile(rows_total: int, dim: int, world: int, scan: str):
    """HBM bytes per launch of the collect-scan kernel from the committed PMC pass (profiles/, FETCH_SIZE
    x2 gfx950 correction + WRITE_SIZE, separate --pmc runs).  bench.py cannot collect PMCs itself, so
    this is only reported when a committed profile matches the workload being run."""
    tag = {(384, "bf16"): "scan16", (384, "f32"): "scan", (768, "bf16"): "scan16_768", (384, "i8"): "scan8",
           (768, "i8"): "scan8_768"}.get((dim, scan))
    if world != 1 or rows_total != 10_000_000 or tag is None:
        return None
    for rnd in ("r6", "r5", "r4", "r3", "r2", "r1"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_{tag}_traffic.json")
        if os.path.exists(path):
            try:
                with open(path) as f:
                    traffic_from_profile.source = os.path.relpath(path, ROOT)
                    return float(json.load(f)["traffic_bytes_per_launch"])
            except Exception:
                return None
    return None


traffic_from_profile.source = None


def query_latency_leg(idx, k: int, calls: int = 200):
    """What one search request costs end to end on the device side of the host API (handlers.rs:61-81: embed the query
    text, search the collection): `encode` of ONE short query (host ids in, host embedding out: mx_encoder_encode) followed by
    `search` of that vector (host pointers: mx_index_search) against the headline corpus, per call, p50 / p99 over `calls`
    calls -- for the reference's two MiniLM models (all-MiniLM-L12-v2 is its default, embedding.rs:67)."""
    from memex_amd import weights as W
    from memex_amd.encoder import Encoder
    out = {"calls": calls, "query_tokens": 16, "k": k,
           "note": "host API both ways; the search leg is one query against the headline corpus (scan at the stream's rate)"}
    rng = np.random.default_rng(5)
    for name, cfg in (("all-MiniLM-L6-v2", W.ALL_MINILM_L6_V2), ("all-MiniLM-L12-v2", W.ALL_MINILM_L12_V2)):
        enc = Encoder(cfg, W.pack_weights(W.synthetic_weights(cfg, 0), cfg))
        ids = rng.integers(1000, cfg.vocab, size=(1, 16)).astype(np.int32)
        lens = np.array([16], dtype=np.int32)
        for _ in range(10):
            v = enc.encode(ids, lens)
            idx.search(v, k)
        te, ts = [], []
        for _ in range(calls):
            t0 = time.perf_counter()
            v = enc.encode(ids, lens)
            t1 = time.perf_counter()
            idx.search(v, k)
            t2 = time.perf_counter()
            te.append((t1 - t0) * 1e3)
            ts.append((t2 - t1) * 1e3)
        enc.close()
        te, ts = np.asarray(te), np.asarray(ts)
        out[name] = {"encode_ms_p50": float(np.percentile(te, 50)), "encode_ms_p99": float(np.percentile(te, 99)),
                     "search_ms_p50": float(np.percentile(ts, 50)), "search_ms_p99": float(np.percentile(ts, 99)),
                     "total_ms_p50": float(np.percentile(te + ts, 50)), "total_ms_p99": float(np.percentile(te + ts, 99))}
    # the third model segment_text accepts (embedding.rs:159) is 768 wide: its embedding does not fit the headline corpus, encode only
    cfg = W.ALL_DISTILROBERTA_V1
    enc = Encoder(cfg, W.pack_weights(W.synthetic_weights(cfg, 0), cfg))
    ids = rng.integers(1000, cfg.vocab, size=(1, 16)).astype(np.int32)
    lens = np.array([16], dtype=np.int32)
    for _ in range(10):
        enc.encode(ids, lens)
    te = []
    for _ in range(calls):
        t0 = time.perf_counter()
        enc.encode(ids, lens)
        te.append((time.perf_counter() - t0) * 1e3)
    enc.close()
    out["all-distilroberta-v1"] = {"encode_ms_p50": float(np.percentile(te, 50)), "encode_ms_p99": float(np.percentile(te, 99))}
    return out


class SearchBuffers:
    def __init__(self, batch: int, k: int):
        import torch
        from memex_amd.index import packed_result_block
        # ids and dists live in one block so that the N>1 exchange is ONE all-gather (B*k*12 bytes per rank)
        self.block, self.ids, self.dists = packed_result_block(batch, k, "cuda")
        self.scores = torch.zeros((batch, k), device="cuda", dtype=torch.float32)
        self.nf = torch.zeros((batch,), device="cuda", dtype=torch.int32)


def concurrent_callers_leg(idx, q, k: int, dim: int, data: str, calls: int, threads: int = 4):
    """Several request threads, each issuing 256-query mx_index_search calls (host pointers) against the headline corpus --
    the reference's API handlers run on a multi-threaded runtime (handlers.rs:55-109).  While one combined batch is on
    the GPU the other callers queue up and the next batch takes up to 512 queries, which the int8 scan serves in one pass:
    two batches' worth of queries share one sample / theta / collect / finish sequence and the host turn-around hides under
    the next callers' queueing.  Reported: aggregate queries/s, scan passes per call (0.5 = every pass carried two calls),
    and that every caller got its own single-caller answers."""
    import threading
    import torch
    qs = [q.cpu().numpy()] + [make_queries(256, dim, data, seed=7000 + t).cpu().numpy() for t in range(1, threads)]
    want = [idx.search(x, k)[0] for x in qs]                      # single-caller answers
    ok = [True] * threads
    start = threading.Barrier(threads + 1)

    def worker(t):
        start.wait()
        for _ in range(calls):
            ids = idx.search(qs[t], k)[0]
            ok[t] = ok[t] and bool((ids == want[t]).all())

    for _ in range(3):
        idx.search(qs[0], k)
    th = [threading.Thread(target=worker, args=(t,)) for t in range(threads)]
    for x in th:
        x.start()
    idx.reset_stats()
    idx.set_profiling(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    start.wait()
    for x in th:
        x.join()
    dt = time.perf_counter() - t0
    st = idx.stats()
    idx.set_profiling(False)
    total_calls = threads * calls
    return {"workload": f"{threads} caller threads x {calls} calls of 256 queries, mx_index_search (host pointers), top-{k}",
            "value": 256 * total_calls / dt, "unit": "queries/s", "ms_per_call": dt / total_calls * 1e3,
            "scan_passes_per_call": st.scan_launches / max(1, total_calls), "retry_queries": int(st.retry_queries),
            "fallback_queries": int(st.fallback_queries), "answers_equal_single_caller": all(ok)}


def roofline_of(st, scan: str, dim: int, batch: int, rows_total: int, world: int):
    scan_s = st.scan_ms / 1e3
    achieved = (st.scan_bytes / scan_s / 1e9) if scan_s > 0 else 0.0
    launches = max(1, st.scan_launches)
    elem = {"i8": 1, "bf16": 2, "f32": 4}[scan]
    mfma_peak = MFMA_PEAK_I8_TOPS if scan == "i8" else MFMA_PEAK_TFLOPS
    tflops = (2.0 * batch * (st.scan_bytes / elem) / scan_s / 1e12) if scan_s > 0 else 0.0
    kc = (dim + 127) // 128
    return {
        "bound": "hbm",
        "kernel": f"mx::scan8_kernel<{kc}, 1, {2 if batch > 256 else 1}, false> (int8 filter copy, collect launch; <..., true> on a centred copy)" if scan == "i8"
                  else f"mx::scan16_kernel<{kc},1> (bf16 filter copy, collect launch)" if scan == "bf16"
                  else f"mx::scan_kernel<{kc},1> (f32 rows, collect launch)",
        "achieved": achieved,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        "bytes_per_launch": st.scan_bytes / launches,
        "ms_per_launch": st.scan_ms / launches,
        # SURVEY 8(d) counts N*D*4 bytes per pass whatever the kernel streams; for the filter-copy scans those bytes are
        # NOT read by this launch (the f32 rows are only touched by the rescoring of a few hundred rows per query), so this
        # figure may exceed 1: it is the throughput expressed in the f32-row scan's units, not a bandwidth
        "frac_f32_bytes_equivalent": ((st.scan_bytes / elem * 4 / scan_s / 1e9) / HBM_PEAK_GBS) if scan_s > 0 else 0.0,
        "frac_f32_bytes_equivalent_note": "N*D*4 bytes / launch time / 8 TB/s -- bytes not read when scan != f32",
        "traffic": traffic_from_profile(rows_total, dim, world, scan),
        "traffic_source": traffic_from_profile.source,   # a committed PMC pass of the same command, not this run (bench.py cannot collect PMCs)
        "mfma_tflops": tflops,
        "mfma_frac": tflops / mfma_peak,
    }


MIN_SECONDS = 0.0   # set from --min-seconds in main()


def timed_steps(idx, step, fence, warmup: int, steps: int, world: int):
    """warm-up, settle (untimed steps until MIN_SECONDS of continuous work precede the measurement: the chip's clock under
    its power cap takes most of a second to settle), then EXACTLY `steps` timed steps between two fences."""
    import torch
    import torch.distributed as dist
    fence()
    t0 = time.perf_counter()
    for _ in range(max(1, warmup)):
        step()
    fence()
    per = (time.perf_counter() - t0) / max(1, warmup)
    settle = int(MIN_SECONDS / per) + 1 if MIN_SECONDS > 0 and per > 0 else 0
    if world > 1:  # every rank must run the same number of steps (collectives inside)
        t = torch.tensor([settle], device="cuda", dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        settle = int(t.item())
    for _ in range(min(settle, 100_000)):
        step()
    idx.reset_stats()
    idx.set_profiling(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    st = idx.stats()
    idx.set_profiling(False)
    timed_steps.last_settle = settle
    return dt, st


def scan_of(st) -> str:
    """What the index's scan streams now (mx_index_stats.filter_kind): the library chooses unless told."""
    return {0: "f32", 2: "i8", 3: "bf16"}[int(st.filter_kind)]


def leg_report(st, dt: float, steps: int, workload: str, dim: int, batch: int, rows: int):
    return {"workload": workload, "value": batch * steps / dt, "unit": "queries/s", "steps": steps,
            "ms_per_step": dt / steps * 1e3, "candidates_per_query": st.candidates / max(1, st.queries),
            "retry_queries": int(st.retry_queries), "fallback_queries": int(st.fallback_queries),
            "approx_err_bound": st.approx_err_bound, "scan": scan_of(st), "filter_demotions": int(st.filter_demotions),
            "filter_centred": int(st.filter_centred),
            "ms_outside_collect_launch": dt / steps * 1e3 - st.scan_ms / max(1, st.scan_launches),
            "roofline": roofline_of(st, scan_of(st), dim, batch, rows, 1)}


def side_leg(rows: int, dim: int, batch: int, k: int, steps: int, data: str):
    """A second workload on a fresh index, N = 1: same measurement as the headline, fewer steps."""
    import torch
    from memex_amd.index import FlatIndex
    idx = FlatIndex(dim, key=None, device=0)
    idx.reserve(rows)
    fill_index(idx, rows, dim, 0, rows, data)
    q = make_queries(batch, dim, data)
    bufs = SearchBuffers(batch, k)
    torch.cuda.synchronize()

    def step():
        idx.search_device(q, k, bufs.ids, bufs.scores, bufs.dists, bufs.nf)
    dt, st = timed_steps(idx, step, torch.cuda.synchronize, 3, steps, 1)
    out = leg_report(st, dt, steps, f"{rows}x{dim} f32 corpus ({data}), query batch {batch}, top-{k}", dim, batch, rows)
    idx.close()
    del idx, q, bufs
    torch.cuda.empty_cache()
    return out


def sharded_one_device_leg(rows: int, dim: int, batch: int, k: int, steps: int, shards: int):
    """(ms per step, ms of it in the serial tail: mx_index_stats.exchange_ms) of the in-library sharded index with `shards`
    logical shards, all on device 0 (a wiring + overhead probe)."""
    import torch
    from memex_amd.index import FlatIndex
    idx = FlatIndex(dim, key=None, device=0, devices=[0] * shards)
    idx.reserve(rows)
    fill_index(idx, rows, dim, 0, rows, "gaussian")
    q = make_queries(batch, dim, "gaussian")
    bufs = SearchBuffers(batch, k)
    torch.cuda.synchronize()

    def step():
        idx.search_device(q, k, bufs.ids, bufs.scores, bufs.dists, bufs.nf)
    dt, st = timed_steps(idx, step, torch.cuda.synchronize, 3, steps, 1)
    idx.close()
    del idx, q, bufs
    torch.cuda.empty_cache()
    return dt / steps * 1e3, st.exchange_ms / steps


def _rounded(x, sig: int = 5):
    """floats to `sig` significant digits, recursively: keeps the one stdout line short"""
    if isinstance(x, float):
        return float(f"{x:.{sig}g}") if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _rounded(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_rounded(v, sig) for v in x]
    return x


def _encoder_claim(leg, tokens: int):
    """What a reader needs to recompute an encoder roofline fraction: chunks/s (wall clock, host ids in -> host embeddings out) x
    GFLOP per chunk / 2.5 PFLOP/s = frac_wall; `frac` itself uses the GPU time of the same chunks (HIP events on the encoder's stream)."""
    if not leg or "value" not in leg:
        return leg
    g = leg["gflop_per_chunk"]
    rep = max(1, int(leg.get("replicas", 1)))  # N > 1: `chunks_per_s` is the sum over the replicas, `frac_wall` divides by their summed peak;
    #                                            `achieved` / `frac` / `gpu_only_chunks_per_s` are ONE replica's (rank 0, HIP events on its stream)
    return {"bound": "mfma", "unit": "TFLOP/s", "peak": MFMA_PEAK_TFLOPS, "achieved": leg["roofline"]["achieved"], "frac": leg["roofline"]["frac"],
            "chunks_per_s": leg["value"], "replicas": rep, "tokens_per_chunk": tokens, "gflop_per_chunk": g,
            "frac_wall": leg["value"] * g / 1e3 / (MFMA_PEAK_TFLOPS * rep),
            "gpu_only_chunks_per_s": leg["gpu_only_chunks_per_s"], "chunks": leg["chunks_per_gpu"],
            "ragged_frac": (leg.get("ragged") or {}).get("frac")}


def _side_summary(name, leg):
    """the few numbers of a side leg that go on the stdout line (its full report: --sides-out)"""
    if not isinstance(leg, dict):
        return leg
    if "error" in leg:
        return {"error": leg["error"][:120]}
    keep = ("value", "unit", "ms_per_step", "scan", "filter_demotions", "filter_centred", "retry_queries", "fallback_queries",
            "ms_outside_collect_launch", "predicted_n8_qps", "search_value", "embed_segments_per_s", "embed_mfma_frac", "windows_per_s",
            "ids_equal_exact_path", "candidates_per_query")
    out = {k: leg[k] for k in keep if k in leg}
    if isinstance(leg.get("roofline"), dict):
        out["frac"] = leg["roofline"].get("frac")
        out["ms_per_launch"] = leg["roofline"].get("ms_per_launch")
    if name == "query_latency":
        out = {k: v for k, v in leg.items() if not isinstance(v, (dict, list))} or {k: (v.get("p50_ms") if isinstance(v, dict) else v) for k, v in leg.items()}
    if name == "small_batches":
        out = {k: {"ms_per_call": v.get("ms_per_call"), "queries_per_s": v.get("queries_per_s")} for k, v in leg.items()}
    return out


def run(a):
    global MIN_SECONDS
    MIN_SECONDS = max(0.0, a.min_seconds)
    import torch
    import torch.distributed as dist

    from memex_amd import _lib
    from memex_amd.index import FlatIndex, merge_topk_packed_device

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    laps, t_lap = {}, [time.perf_counter()]

    def lap(name):  # wall seconds per stage of this script (corpus builds and CPU baselines included): `wall_s` in the line
        now = time.perf_counter()
        laps[name] = round(laps.get(name, 0.0) + now - t_lap[0], 2)
        t_lap[0] = now
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    # one plain process and N > 1: the in-library sharded index drives all N GPUs (the form the single-process
    # Rust host would run); under torch.distributed.run (WORLD_SIZE set): one process per GPU
    in_library = a.gpus > 1 and world == 1
    if in_library and a.per_process:
        raise SystemExit("--per-process needs: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    shards = a.gpus if in_library else 1
    if os.environ.get("MEMEX_BENCH_TEST_FAIL") == ("in-library" if in_library else "per-process") and a.gpus > 1:
        raise RuntimeError("MEMEX_BENCH_TEST_FAIL: injected failure of this multi-GPU form (wiring test of the fallback)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    # MEMEX_BENCH_ONE_DEVICE=1 is a wiring check for boxes with a single GPU: all ranks / shards share device 0
    # (gloo for the per-process collectives: RCCL refuses two ranks on one device).  Never a benchmark.
    one_device = os.environ.get("MEMEX_BENCH_ONE_DEVICE") == "1"
    if in_library and not one_device and torch.cuda.device_count() < a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but only {torch.cuda.device_count()} device(s) are visible")
    dev = local_rank if (world > 1 and not one_device) else 0
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    if "MEMEX_HIP_SPIN" not in os.environ:
        os.environ["MEMEX_HIP_SPIN"] = "1"  # a benchmark owns its core: poll the completion word (servers sleep by default)

    # ---- corpus shard in HBM (generated on device in blocks; ids are global)
    rows_total = a.rows
    lo = rows_total * rank // world
    hi = rows_total * (rank + 1) // world
    n_local = (hi - lo) // shards
    shard_devs = ([0] * shards if one_device else list(range(shards))) if in_library else None
    idx = FlatIndex(a.dim, key=None, device=dev, devices=shard_devs)
    idx.set_filter_copy({"f32": False, "bf16": "bf16", "i8": "i8"}[a.scan])
    idx.reserve(hi - lo)
    idx.set_id_offset(lo)
    fill_index(idx, rows_total, a.dim, lo, hi, a.data)
    q = make_queries(a.batch, a.dim, a.data)
    k = a.k
    bufs = SearchBuffers(a.batch, k)
    if world > 1:
        g_block = torch.zeros((world, bufs.block.numel()), device="cuda", dtype=torch.uint8)
        m_ids = torch.zeros((a.batch, k), device="cuda", dtype=torch.int64)
        m_dists = torch.zeros((a.batch, k), device="cuda", dtype=torch.float32)
        m_scores = torch.zeros((a.batch, k), device="cuda", dtype=torch.float32)
    torch.cuda.synchronize()

    def step():
        # blocks until results are in HBM; on a sharded index: every shard's scan, the exchange and the merge
        idx.search_device(q, k, bufs.ids, bufs.scores, bufs.dists, bufs.nf)
        if world > 1:
            if one_device:  # gloo: gather through host memory
                parts = [torch.empty_like(bufs.block, device="cpu") for _ in range(world)]
                dist.all_gather(parts, bufs.block.cpu())
                g_block.copy_(torch.stack(parts).to(g_block.device))
            else:
                dist.all_gather_into_tensor(g_block, bufs.block)
            # the merge is enqueued on torch's stream, behind the collective: no host round trip here
            merge_topk_packed_device(dev, g_block, world, a.batch, k, m_ids, m_dists, m_scores,
                                     stream=torch.cuda.current_stream().cuda_stream)

    def fence():
        if world > 1:
            dist.barrier()
        for d in sorted(set(shard_devs or [dev])):
            torch.cuda.synchronize(d)

    dt, st = timed_steps(idx, step, fence, a.warmup, a.steps, world)
    lap("headline (build + timed steps)")
    settle_steps = getattr(timed_steps, "last_settle", 0)
    ids_main = bufs.ids.clone()
    exchange = idx.exchange
    alt = None
    f32_leg = None
    single = world == 1 and not in_library
    if single and a.alt_steps > 0:
        # the same job on the other scan kernels (results must be identical: same certificate, same rescoring): the bf16
        # filter copy, and the f32 rows themselves -- the kernel SURVEY 8(d)'s N*D*4 bytes per pass describe literally
        legs = {}
        for other in [s_ for s_ in ("bf16", "f32") if s_ != a.scan]:
            idx.set_filter_copy("bf16" if other == "bf16" else False)
            dt2, st2 = timed_steps(idx, step, fence, 3, a.alt_steps, world)
            legs[other] = {"scan": other, "value": a.batch * a.alt_steps / dt2, "unit": "queries/s", "steps": a.alt_steps,
                           "ms_per_step": dt2 / a.alt_steps * 1e3, "ids_equal_main_run": bool(torch.equal(bufs.ids, ids_main)),
                           "roofline": roofline_of(st2, other, a.dim, a.batch, rows_total, world)}
        f32_leg = legs.pop("f32", None)
        alt = next(iter(legs.values()), None)
        idx.set_filter_copy({"f32": False, "bf16": "bf16", "i8": "i8"}[a.scan])
        step()
    lap("other_scan + f32_rows")
    host_api = None
    if single and a.side_steps > 0:
        # the entry point the Rust shim binds: host pointers in and out (queries H2D, results D2H inside the step)
        qh = q.cpu().numpy()
        dt3, st3 = timed_steps(idx, lambda: idx.search(qh, k), fence, 3, a.side_steps, world)
        host_api = leg_report(st3, dt3, a.side_steps, f"{rows_total}x{a.dim} f32 corpus ({a.data}), query batch {a.batch}, "
                              f"top-{k}, mx_index_search (host pointers)", a.dim, a.batch, rows_total)

    callers = None
    if single and a.side_steps > 0 and a.batch == 256:
        callers = concurrent_callers_leg(idx, q, k, a.dim, a.data, a.side_steps)
        lap("host_api + concurrent_callers")

    small = None
    small_steps = a.side_steps if a.small_steps is None else a.small_steps
    if single and small_steps > 0:
        # small batches through the same entry point: a lone query (what the reference's trait issues) and 32 of them.
        # Waves of the scan without live queries skip their MFMAs, so these run at the stream's rate, not at the
        # power-capped rate of a full batch.
        small = {}
        for nb in (1, 32):
            qs = q[:nb].contiguous()
            sb = SearchBuffers(nb, k)
            dts, sts = timed_steps(idx, lambda: idx.search_device(qs, k, sb.ids, sb.scores, sb.dists, sb.nf), fence, 3, small_steps, world)
            rf = roofline_of(sts, a.scan, a.dim, nb, rows_total, 1)
            small[f"batch_{nb}"] = {"ms_per_call": dts / small_steps * 1e3, "queries_per_s": nb * small_steps / dts,
                                    "collect_ms": rf["ms_per_launch"], "collect_GBps": rf["achieved"], "collect_hbm_frac": rf["frac"],
                                    "ids_equal_full_batch": bool(torch.equal(sb.ids, ids_main[:nb]))}
        step()
        # more than 256 queries in one call: the int8 scan serves 512 per pass (two query groups per wave) up to 512 dims
        if a.batch == 256:
            qb_ = torch.cat([q, make_queries(256, a.dim, a.data, seed=9876)]).contiguous()
            bb = SearchBuffers(512, k)
            dtb, stb = timed_steps(idx, lambda: idx.search_device(qb_, k, bb.ids, bb.scores, bb.dists, bb.nf), fence, 3, small_steps, world)
            rfb = roofline_of(stb, a.scan, a.dim, 512, rows_total, 1)
            small["batch_512"] = {"ms_per_call": dtb / small_steps * 1e3, "queries_per_s": 512 * small_steps / dtb,
                                  "collect_ms": rfb["ms_per_launch"], "passes_per_call": stb.scan_launches / max(1, small_steps),
                                  "ids_equal_full_batch": bool(torch.equal(bb.ids[:256], ids_main))}
            del qb_, bb

    qlat = None
    if single and small_steps > 0:
        qlat = query_latency_leg(idx, k)
        lap("small_batches + query_latency")

    # ---- recall@10 against the all-f64 EXACT path on a few queries (oracle-level parity at smaller
    # sizes and the oracle-based 10M check live in tests/; the EXACT path is itself oracle-checked there)
    recall = recall_exact_order = merged_ok = None
    if rank == 0 and a.recall_queries > 0:
        nq = min(a.recall_queries, a.batch)
        final_ids = (m_ids if world > 1 else bufs.ids)[:nq].clone()
        if world == 1:  # (on the in-library sharded index too: the EXACT mode reaches every shard)
            e = SearchBuffers(nq, k)
            idx.set_search_mode(_lib.MX_SEARCH_EXACT)
            idx.search_device(q[:nq].contiguous(), k, e.ids, e.scores, e.dists, e.nf)
            idx.set_search_mode(_lib.MX_SEARCH_AUTO)
            hit = sum(len(set(final_ids[b].tolist()) & set(e.ids[b].tolist())) for b in range(nq))
            recall = hit / float(nq * k)
            recall_exact_order = bool(torch.equal(final_ids, e.ids))
        else:
            # N > 1: no rank holds the whole corpus; check the merge's invariants instead (lists ordered by
            # (dist, id), ids unique and inside the corpus, every list full)
            md, mi = m_dists.cpu(), m_ids.cpu()
            merged_ok = bool((md[:, 1:] >= md[:, :-1]).all()) and bool(((mi >= 1) & (mi <= rows_total)).all()) and \
                all(len(set(r.tolist())) == k for r in mi)
    if world > 1:
        dist.barrier()
    idx.close()
    del idx
    torch.cuda.empty_cache()

    sides = {}
    if single and a.side_steps > 0:
        sides["host_api"] = host_api
        sides["concurrent_callers"] = callers
        sides["small_batches"] = small
        sides["query_latency"] = qlat
        other_data = "clustered" if a.data == "gaussian" else "gaussian"
        sides[other_data] = side_leg(rows_total, a.dim, a.batch, k, a.side_steps, other_data)
        if a.data == "gaussian":  # embedding-like rows (decaying spectrum + common mean direction), the library's own choice of copy
            sides["anisotropic"] = side_leg(rows_total, a.dim, a.batch, k, a.side_steps, "anisotropic")
        sides["cfg4_shard_10Mx768"] = side_leg(10_000_000, 768, a.batch, k, a.side_steps, "gaussian")
        lap("clustered + anisotropic + cfg4 shard")
        if a.shard_legs:
            # what each of 8 GPUs runs per step when configs[2] / a 10M x 768 corpus is sharded 8 ways: the whole per-step fixed
            # cost (prep, sample, theta, finish) against 1/8 of the scan -- a one-GPU proxy for strong scaling (no exchange)
            for nm, dm in (("shard_1p25Mx384", 384), ("shard_1p25Mx768", 768)):
                leg = side_leg(1_250_000, dm, a.batch, k, a.side_steps, "gaussian")
                leg["predicted_n8_qps_no_exchange"] = a.batch / leg["ms_per_step"] * 1e3
                # the exchange + merge term, timed on this one device: the in-library sharded index with EIGHT logical shards
                # on device 0 answers the same batch (its shards take turns on the GPU, each step host-synchronised like the
                # plain leg, then peer copies of the eight [ids | dists] blocks and merge_kernel); what it takes beyond eight
                # plain shard steps is what a step of the 8-GPU job adds to one shard step -- with peer copies where the real
                # job has ONE RCCL all-gather of 8 x 30 KB over xGMI (tens of us; unmeasured: no multi-GPU node yet)
                try:
                    t8, tail = sharded_one_device_leg(8 * 1_250_000, dm, a.batch, k, a.side_steps, 8)
                    # a shard's own share of a sharded step (its scan pipeline + the query copy in + its block copied out) and
                    # the serial tail behind the last shard (mx_index_stats.exchange_ms: merge_kernel, n_found, synchronise)
                    shard_local = max(leg["ms_per_step"], (t8 - tail) / 8.0)
                    leg["eight_logical_shards_ms_per_step"] = t8
                    leg["shard_local_ms"] = shard_local
                    leg["serial_tail_ms"] = tail
                    leg["exchange_and_merge_ms"] = shard_local - leg["ms_per_step"] + tail
                    leg["predicted_n8_qps"] = a.batch / (shard_local + tail) * 1e3
                    leg["note"] = ("predicted_n8_qps = batch / (one shard's share of a sharded step + the serial tail), both timed with 8 "
                                   "logical shards on one device (peer copies instead of the RCCL all-gather); a prediction, not a measurement")
                except Exception as e:  # noqa: BLE001 -- a side leg must not fail the bench
                    leg["exchange_and_merge_error"] = repr(e)[:300]
                sides[nm] = leg
            lap("shard legs")
        if a.enc_like_rows > 0:
            sides["enc_like_10M"] = enc_like_leg(a.enc_like_rows, 100_000, a.batch, k, a.side_steps)
            lap("enc_like_10M")
        if a.cfg2_segments > 0:
            sides["cfg2"] = cfg2_leg(a.cfg2_segments, a.batch, k, a.side_steps)
            lap("cfg2")
        if a.precise_chunks > 0:
            try:
                sides["ingest_bf16x3"] = precise_ingest_leg(a.precise_chunks, a.precise_chunks // 4)
            except Exception as e:  # noqa: BLE001 -- a side leg must not fail the bench
                sides["ingest_bf16x3"] = {"error": repr(e)[:300]}
            lap("ingest_bf16x3")
        if a.text_docs > 0:
            try:
                sides["text_ingest"] = text_ingest_leg(a.text_docs, cpu_too=not a.no_cpu_baseline)
            except Exception as e:  # noqa: BLE001 -- a side leg must not fail the bench
                sides["text_ingest"] = {"error": repr(e)[:300]}
            lap("text_ingest (incl. its CPU baseline)")
    ingest = None
    if a.ingest_chunks > 0:
        ingest = ingest_leg(a.ingest_chunks, dev, world, not a.no_cpu_baseline, devices=shard_devs if in_library and not one_device else None)
    ingest_bge = ingest_short = None
    if single and a.bge_chunks > 0:
        ingest_bge = ingest_leg(a.bge_chunks, dev, 1, not a.no_cpu_baseline, model="bge-base-en")
    if single and a.short_seqs > 0:
        # the reference's default model at ITS window (embedding.rs:64-73: all-MiniLM-L12-v2, max_seq_length 128)
        ingest_short = ingest_leg(a.short_seqs, dev, 1, False, model="all-MiniLM-L12-v2", seq=128)
    lap("ingest + ingest_bge_base + 128-token leg (incl. their CPU baselines)")

    if rank == 0:
        n_gpus = world * shards
        roof = roofline_of(st, a.scan, a.dim, a.batch, n_local, n_gpus)
        out = {
            "metric": "queries/sec, exact cosine top-10 (recall@10 = 1.0) on 10M x 384-d f32",
            "value": a.batch * a.steps / dt,
            "unit": "queries/s",
            "n_gpus": n_gpus,
            "steps": a.steps,
            "warmup": a.warmup,
            "settle_steps": settle_steps,   # untimed steps of the same work in front of the timed region (--min-seconds)
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": {"i8": "f32 corpus; int8 MFMA filter (exact integer sums, measured-residual certificate)",
                      "bf16": "f32 corpus; bf16 MFMA filter", "f32": "f32 corpus; bf16 MFMA filter on rows rounded in flight"}[a.scan]
                     + " -> f32 rescoring -> f64 DistCosine on the survivors (results bit-identical to all-f64)",
            "scan": a.scan,
            "filter_copy_bytes": int(st.filter_copy_bytes),
            "data": "synthetic" if a.data == "gaussian" else f"synthetic ({a.data})",
            "config": {"workload": f"{rows_total}x{a.dim} f32 corpus in HBM ({a.data}), query batch {a.batch}, top-{k}",
                       "rows_per_gpu": n_local,
                       "wait": "poll" if os.environ.get("MEMEX_HIP_SPIN") == "1" else "sleep",
                       "parallelism": (f"row-shard x{shards} inside one process (mx_index_open_sharded)" if in_library else
                                       f"row-shard x{world}, one process per GPU" if world > 1 else "single GPU")},
            "rccl_ranks": n_gpus if (world > 1 and not one_device) or exchange == "rccl" else 0,
            "exchange": exchange if in_library else ("rccl" if world > 1 and not one_device else ("gloo" if world > 1 else "none")),
            "recall_at_10": recall,
            "recall_note": "vs the library's own EXACT path (all-f64) on a few queries; the oracle-based check at this size is tests/test_search_fullsize_gpu.py",
            "ids_equal_exact_path": recall_exact_order,
            "merged_lists_ok": merged_ok,
            "fallback_queries": int(st.fallback_queries),
            "retry_queries": int(st.retry_queries),
            "candidates_per_query": st.candidates / max(1, st.queries),
            "approx_err_bound": st.approx_err_bound,
            "ms_outside_collect_launch": dt / a.steps * 1e3 - roof["ms_per_launch"],
            "roofline": roof,
        }
        # ---- every roofline claim where the driver records it: nested under `roofline`
        if f32_leg is not None:
            fr = f32_leg["roofline"]
            roof["section_8d_kernel"] = {"kernel": fr["kernel"], "what": "the scan over the f32 rows themselves: SURVEY 8(d)'s N*D*4 bytes per pass, literally "
                                         "(--scan f32; not the default: the int8 filter copy answers the same queries 2.4x faster)",
                                         "bytes_per_launch": fr["bytes_per_launch"], "ms_per_launch": fr["ms_per_launch"], "achieved": fr["achieved"],
                                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fr["frac"], "queries_per_s": f32_leg["value"],
                                         "ms_per_step": f32_leg["ms_per_step"], "ids_equal_main_run": f32_leg["ids_equal_main_run"]}
        if alt is not None:
            ar = alt["roofline"]
            roof["other_filter_copy"] = {"scan": alt["scan"], "kernel": ar["kernel"], "bytes_per_launch": ar["bytes_per_launch"],
                                         "ms_per_launch": ar["ms_per_launch"], "frac": ar["frac"], "queries_per_s": alt["value"],
                                         "ids_equal_main_run": alt["ids_equal_main_run"]}
        if ingest is not None:
            roof["encoder_minilm"] = _encoder_claim(ingest, 512)
        if ingest_bge is not None:
            roof["encoder_bge"] = _encoder_claim(ingest_bge, 512)
        if ingest_short is not None:
            roof["encoder_minilm_128tok"] = dict(_encoder_claim(ingest_short, 128), model="all-MiniLM-L12-v2 shape, 128-token sequences (the reference default's own window)")
        if isinstance(sides.get("ingest_bf16x3"), dict):
            roof["encoder_split_modes"] = sides.pop("ingest_bf16x3")
        if single and not a.no_cpu_baseline:
            cb = cpu_bruteforce(a.dim, a.batch, k, rows_total, a.cpu_seconds)
            if a.hnsw_rows > 0:
                try:
                    cb["hnsw"] = cpu_hnsw(a.dim, a.batch, k, a.hnsw_rows)
                    # the same at up to --hnsw-big-rows on the clustered corpus (the one HNSW has a recall on), cut to the rows whose
                    # parallel build fits the time bound (extrapolated ~ n log n from the run above; a 1M build is 5-8 minutes on 128 cores)
                    if a.hnsw_big_rows > a.hnsw_rows and a.hnsw_big_seconds > 0:
                        b0 = next(r["build_s"] for r in cb["hnsw"]["runs"] if r["data"] == "clustered")
                        import math
                        n = a.hnsw_big_rows
                        while n > a.hnsw_rows and 1.25 * b0 * (n / a.hnsw_rows) * math.log(n) / math.log(a.hnsw_rows) > a.hnsw_big_seconds:
                            n = int(n * 0.9)
                        if n > 1.5 * a.hnsw_rows:
                            big = cpu_hnsw(a.dim, a.batch, k, n, corpora=("clustered",))
                            big["note"] = f"target {a.hnsw_big_rows} rows, cut to {n} so that the build fits {a.hnsw_big_seconds:.0f} s"
                            cb["hnsw_big"] = big
                except Exception as e:  # the baseline must never fail the bench
                    cb["hnsw_big" if "hnsw" in cb else "hnsw"] = {"error": repr(e)[:300]}
            if ingest is not None:
                cb["encoder_minilm"] = ingest.get("cpu_baseline")
            if ingest_bge is not None:
                cb["encoder_bge"] = ingest_bge.get("cpu_baseline")
            out["cpu_baseline"] = cb
            lap("cpu_baseline (brute force + HNSW)")
        # ---- the side legs: full reports to a second file and stderr, a compact summary on the line
        if ingest is not None:
            sides["ingest"] = ingest
        if ingest_bge is not None:
            sides["ingest_bge_base"] = ingest_bge
        if ingest_short is not None:
            sides["ingest_minilm_l12_128tok"] = ingest_short
        if alt is not None:
            sides["other_scan"] = alt
        if f32_leg is not None:
            sides["f32_rows"] = f32_leg
        if sides:
            if a.sides_out:
                path = a.sides_out
            elif os.path.isdir(os.path.join(ROOT, "gpurun_out")):
                path = os.path.join(ROOT, "gpurun_out", "bench_sides.json")
            else:
                path = os.path.join(os.getcwd(), "bench_sides.json")
            try:
                with open(path, "w") as f:
                    json.dump(sides, f, indent=1)
                out["sides_file"] = os.path.relpath(path, ROOT)
            except OSError as e:
                out["sides_file"] = f"not written: {e}"
            print("bench.py side legs: " + json.dumps(sides), file=sys.stderr, flush=True)
            out["sides"] = {nm: _side_summary(nm, leg) for nm, leg in sides.items()
                            if nm not in ("ingest", "ingest_bge_base", "ingest_minilm_l12_128tok", "other_scan", "f32_rows")}
        out["wall_s"] = laps
        print(json.dumps(_rounded(out)), flush=True)
    if world > 1:
        dist.destroy_process_group()


def _other_form(a, err: str):
    """N > 1 and this form failed: run the OTHER multi-GPU form in a child process and relay its line, so that a first
    8-GPU run yields a measurement whichever form the node supports.  Returns the child's return code."""
    import subprocess
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args = [f"--{k.replace('_', '-')}={v}" for k, v in vars(a).items()
            if k not in ("per_process", "no_fallback", "no_cpu_baseline", "small_steps") and v is not None]
    args += ["--no-fallback"] + (["--no-cpu-baseline"] if a.no_cpu_baseline else [])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                            "ROLE_RANK", "ROLE_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                            "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS")}
    if world > 1:  # the per-process form failed -> one plain process driving every GPU through the in-library sharded index
        cmd = [sys.executable, os.path.abspath(__file__)] + args
    else:          # the in-library form failed -> one process per GPU
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(29500 + os.getpid() % 2000), os.path.abspath(__file__)] + args + ["--per-process"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True)
    line = next((ln for ln in reversed(r.stdout.splitlines()) if ln.startswith("{")), None)
    if line is None:
        print(json.dumps({"metric": "queries/sec, exact cosine top-10 (recall@10 = 1.0) on 10M x 384-d f32", "value": None,
                          "unit": "queries/s", "n_gpus": a.gpus, "steps": a.steps, "warmup": a.warmup, "higher_is_better": True,
                          "error": err, "fallback_error": (r.stderr or r.stdout)[-800:]}), flush=True)
        return r.returncode or 1
    out = json.loads(line)
    out["fallback_from"] = "one process per GPU (torch.distributed.run)" if world > 1 else "in-library sharded index (one process)"
    out["error"] = err
    print(json.dumps(out), flush=True)
    return 0


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    try:
        run(a)
    except BaseException as e:  # noqa: BLE001 -- a multi-GPU bench must leave a line (and try the other form) whatever broke
        if isinstance(e, KeyboardInterrupt) or a.gpus <= 1 or a.no_fallback:
            raise
        import traceback
        err = f"{type(e).__name__}: {e}"[:600]
        traceback.print_exc()
        if rank != 0:
            raise SystemExit(0)  # (not an error code: the launcher would tear rank 0 down in the middle of its fallback)
        raise SystemExit(_other_form(a, err))


if __name__ == "__main__":
    main()
