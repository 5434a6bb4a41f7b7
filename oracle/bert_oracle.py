"""oracle/bert_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

NumPy (float64) restatement of the sentence-encoder forward that memex runs through
``model.encode(&segments)`` (lib/libmemex/src/llm/embedding.rs:109).  The arithmetic itself lives in
un-vendored crates -- rust-bert 0.21.0 (Cargo.lock:3467-3469) on tch 0.13.0 / libtorch -- so this
follows the published BERT / sentence-transformers semantics those crates implement
(SURVEY.md App. A.1):

* embeddings = word[id] + position[pos_offset + t] + token_type[0]  -> LayerNorm(eps)
  (pos_offset = 0 for BERT; 2 for RoBERTa-style checkpoints: positions start at padding_idx + 1)
* per layer: Q,K,V = x W^T + b ; scores = QK^T/sqrt(d_head) + (1-mask)*(-10000) ; softmax ; PV ;
  dense ; +residual ; LayerNorm ; dense(H->F) ; GELU(erf) ; dense(F->H) ; +residual ; LayerNorm
* pooling: masked mean  sum(h*m)/max(sum(m), 1e-9)   (all-MiniLM-*)   or CLS (bge-*)
* optional L2 normalise  x / max(||x||_2, 1e-12)

PARITY UNPINNED for embedding *values*: no reference test asserts any (the only encoder test,
embedding.rs:204-217, checks a padded token count and needs the network).  The restatement is
cross-checked here against ``transformers.BertModel`` (tests/golden/make_encoder_golden.py, run in
the dev container only) and the resulting vectors are committed under tests/golden/.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.
"""
from __future__ import annotations

import math

import numpy as np

try:  # scipy is present in the image; keep a fallback so the oracle has no hard dependency
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover
    _erf = np.vectorize(math.erf)


def layer_norm(x: np.ndarray, g: np.ndarray, b: np.ndarray, eps: float) -> np.ndarray:
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * g + b


def gelu_erf(x: np.ndarray) -> np.ndarray:
    return 0.5 * x * (1.0 + _erf(x / math.sqrt(2.0)))


def encode(weights: dict, cfg: dict, ids: np.ndarray, lens: np.ndarray, dtype=np.float64,
           return_hidden: bool = False):
    """ids: [B,S] int (positions >= lens[b] are padding); returns [B,H] float64.

    ``weights`` uses HF BertModel tensor names (no ``bert.`` prefix); Linear weights are [out,in].
    ``cfg`` keys: layers, hidden, heads, ffn, ln_eps, pooling ('mean'|'cls'), normalize (bool).
    """
    W = {k: np.asarray(v, dtype=dtype) for k, v in weights.items()}  # no copy when already `dtype` (encode_many)
    ids = np.asarray(ids)
    B, S = ids.shape
    H, nh, L = cfg["hidden"], cfg["heads"], cfg["layers"]
    dh = H // nh
    eps = cfg.get("ln_eps", 1e-12)
    lens = np.asarray(lens)
    mask = (np.arange(S)[None, :] < lens[:, None]).astype(dtype)  # [B,S]

    x = (W["embeddings.word_embeddings.weight"][ids]
         + W["embeddings.position_embeddings.weight"][None, cfg.get("pos_offset", 0):cfg.get("pos_offset", 0) + S]
         + W["embeddings.token_type_embeddings.weight"][0][None, None, :])
    x = layer_norm(x, W["embeddings.LayerNorm.weight"], W["embeddings.LayerNorm.bias"], eps)
    add_mask = (1.0 - mask)[:, None, None, :] * -10000.0

    for l in range(L):
        p = f"encoder.layer.{l}."

        def lin(t, name):  # one 2-D GEMM (NumPy's batched matmul of a 3-D operand is ~50x slower)
            w = W[p + name + ".weight"]
            return (t.reshape(-1, t.shape[-1]) @ w.T + W[p + name + ".bias"]).reshape(t.shape[:-1] + (w.shape[0],))

        def heads(t):
            return np.ascontiguousarray(t.reshape(B, S, nh, dh).transpose(0, 2, 1, 3))

        q = heads(lin(x, "attention.self.query"))
        k = heads(lin(x, "attention.self.key"))
        v = heads(lin(x, "attention.self.value"))
        s = q @ np.ascontiguousarray(k.transpose(0, 1, 3, 2)) / math.sqrt(dh) + add_mask
        s = s - s.max(axis=-1, keepdims=True)
        pr = np.exp(s)
        pr = pr / pr.sum(axis=-1, keepdims=True)
        ctx = (pr @ v).transpose(0, 2, 1, 3).reshape(B, S, H)
        a = lin(ctx, "attention.output.dense")
        x = layer_norm(a + x, W[p + "attention.output.LayerNorm.weight"],
                       W[p + "attention.output.LayerNorm.bias"], eps)
        h = gelu_erf(lin(x, "intermediate.dense"))
        o = lin(h, "output.dense")
        x = layer_norm(o + x, W[p + "output.LayerNorm.weight"], W[p + "output.LayerNorm.bias"], eps)

    if cfg.get("pooling", "mean") == "cls":
        pooled = x[:, 0, :]
    else:
        pooled = (x * mask[:, :, None]).sum(axis=1) / np.maximum(mask.sum(axis=1, keepdims=True), 1e-9)
    if cfg.get("normalize", True):
        pooled = pooled / np.maximum(np.linalg.norm(pooled, axis=1, keepdims=True), 1e-12)
    if return_hidden:
        return pooled, x
    return pooled


def encode_many(weights: dict, cfg: dict, ids: np.ndarray, lens: np.ndarray, batch: int = 16, workers: int = 0) -> np.ndarray:
    """:func:`encode` over many sequences: sorted by length, cut into batches trimmed to their longest
    sequence (padding beyond a row's length does not change its result: it is masked), batches spread over
    threads (NumPy releases the GIL inside its kernels).  Same arithmetic per sequence as :func:`encode`."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    ids = np.asarray(ids)
    lens = np.asarray(lens)
    W = {k: np.asarray(v, dtype=np.float64) for k, v in weights.items()}
    order = np.argsort(lens, kind="stable")
    out = np.zeros((ids.shape[0], cfg["hidden"]), dtype=np.float64)
    jobs = [order[i:i + batch] for i in range(0, len(order), batch)]

    def run(sel):
        smax = int(lens[sel].max())
        out[sel] = encode(W, cfg, ids[sel][:, :smax], lens[sel])

    workers = workers or min(32, os.cpu_count() or 1)
    try:
        from threadpoolctl import threadpool_limits
        limit = threadpool_limits(limits=max(1, (os.cpu_count() or 1) // workers))
    except Exception:  # pragma: no cover
        limit = None
    try:
        with ThreadPoolExecutor(max_workers=workers) as ex:
            list(ex.map(run, jobs))
    finally:
        if limit is not None:
            limit.unregister()
    return out
