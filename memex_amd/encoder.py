"""Object wrapper over the ``mx_encoder_*`` C ABI (include/memex_hip.h): the HIP sentence encoder
that replaces the rust-bert model behind ``model.encode(&segments)``
(reference lib/libmemex/src/llm/embedding.rs:99-100,109)."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from ._lib import EncoderCfg, EncoderStats, check, lib
from .weights import EncoderConfig, pack_weights


def _cfg_struct(cfg: EncoderConfig) -> EncoderCfg:
    return EncoderCfg(cfg.layers, cfg.hidden, cfg.heads, cfg.ffn, cfg.vocab, cfg.max_pos, cfg.type_vocab,
                      cfg.ln_eps, _lib.MX_POOL_CLS if cfg.pooling == "cls" else _lib.MX_POOL_MEAN,
                      1 if cfg.normalize else 0, cfg.pos_offset,
                      {"bf16": _lib.MX_PREC_BF16, "bf16x3": _lib.MX_PREC_BF16X3, "mixed": _lib.MX_PREC_MIXED,
                       "mixed1": _lib.MX_PREC_MIXED1}[cfg.precision])


class Encoder:
    def __init__(self, cfg: EncoderConfig, weights, device: int = 0, key: str | None = None):
        """``weights``: packed f32 blob (np.ndarray) or a mapping of HF tensor names (None = attach to
        the resident encoder ``key``).  ``key``: register / attach under that name: every Encoder opened
        with the same key shares ONE resident set of weights (the reference reloads the checkpoint per
        request, handlers.rs:61-63)."""
        self.cfg = cfg
        self.device = device
        c = _cfg_struct(cfg)
        h = ctypes.c_void_p()
        if weights is None:
            check(lib().mx_encoder_open(key.encode() if key else None, ctypes.byref(c), None, 0, device, ctypes.byref(h)))
        else:
            blob = weights if isinstance(weights, np.ndarray) else pack_weights(weights, cfg)
            blob = np.ascontiguousarray(blob, dtype=np.float32)
            check(lib().mx_encoder_open(key.encode() if key else None, ctypes.byref(c), blob.ctypes.data_as(ctypes.c_void_p),
                                        blob.nbytes, device, ctypes.byref(h)))
        self._h = h

    @staticmethod
    def weight_bytes(cfg: EncoderConfig) -> int:
        c = _cfg_struct(cfg)
        return int(lib().mx_encoder_weight_bytes(ctypes.byref(c)))

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().mx_encoder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def encode(self, ids, lens) -> np.ndarray:
        """ids [B,S] int, lens [B] -> [B, hidden] f32 (pooled, L2-normalised per cfg)."""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        B, S = ids.shape
        out = np.zeros((B, self.cfg.hidden), dtype=np.float32)
        check(lib().mx_encoder_encode(self._h, ids.ctypes.data_as(ctypes.c_void_p),
                                      lens.ctypes.data_as(ctypes.c_void_p), B, S,
                                      out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def encode_device(self, ids, lens, out) -> None:
        """Device tensors: ids int32 [B,S], lens int32 [B], out f32 [B,hidden] (blocks until done)."""
        B, S = int(ids.shape[0]), int(ids.shape[1])
        try:  # inputs were produced on torch's current stream (stream contract, include/memex_hip.h)
            import torch
            check(lib().mx_encoder_wait_stream(self._h, ctypes.c_void_p(torch.cuda.current_stream(ids.device).cuda_stream)))
        except ImportError:
            pass
        check(lib().mx_encoder_encode_device(self._h, ctypes.c_void_p(ids.data_ptr()), ctypes.c_void_p(lens.data_ptr()),
                                             B, S, ctypes.c_void_p(out.data_ptr())))

    def set_profiling(self, on: bool) -> None:
        check(lib().mx_encoder_set_profiling(self._h, 1 if on else 0))

    def stats(self) -> EncoderStats:
        s = EncoderStats()
        check(lib().mx_encoder_get_stats(self._h, ctypes.byref(s)))
        return s

    def reset_stats(self) -> None:
        check(lib().mx_encoder_reset_stats(self._h))
