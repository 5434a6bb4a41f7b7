"""ctypes binding of ``libmemex_hip.so`` (C ABI: ``include/memex_hip.h``).

There is deliberately no CPU fallback: if the shared library is missing or a call fails, this
module raises.  ``build()`` compiles the library in-tree with hipcc for gfx950.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmemex_hip.so")
# the same sources built with -DMEMEX_TESTING: two fault-injection hooks (the first RCCL all-gather of a process fails;
# mx_tokenizer_encode_staged throws on request).  Only tests load it -- testing_lib() / use_testing_library() below.
TESTING_LIB_PATH = os.path.join(_HERE, "libmemex_hip_testing.so")
CSRC = os.path.join(_HERE, "csrc")

MX_OK = 0
MX_EINVAL, MX_EDEVICE, MX_EINSERT, MX_ESEARCH, MX_EIO, MX_EUNSUPPORTED, MX_ENOMEM = -1, -2, -3, -4, -5, -6, -7
MX_SEARCH_AUTO, MX_SEARCH_EXACT = 0, 1
MX_CORPUS_F32, MX_CORPUS_BF16 = 0, 1
MX_POOL_MEAN, MX_POOL_CLS = 0, 1
MX_PREC_BF16, MX_PREC_BF16X3, MX_PREC_MIXED, MX_PREC_MIXED1 = 0, 1, 2, 3

# every symbol include/memex_hip.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "mx_last_error", "mx_version", "mx_index_stats_size", "mx_device_count",
    "mx_index_open", "mx_index_open_sharded", "mx_index_n_shards", "mx_index_exchange", "mx_index_wait_stream", "mx_index_close", "mx_index_dim", "mx_index_size", "mx_index_reserve",
    "mx_index_set_id_offset", "mx_index_add", "mx_index_add_device", "mx_index_clear",
    "mx_index_search", "mx_index_search_device", "mx_index_set_search_mode", "mx_index_set_filter_copy", "mx_index_set_corpus_mode", "mx_index_get_rows",
    "mx_index_save", "mx_index_load", "mx_index_has_store", "mx_index_store_info", "mx_index_remove_files",
    "mx_index_set_profiling", "mx_index_get_stats", "mx_index_reset_stats", "mx_topk_merge_device", "mx_topk_merge_packed_device", "mx_topk_merge_packed_async",
    "mx_encoder_cfg_size", "mx_encoder_weight_bytes", "mx_encoder_create", "mx_encoder_open", "mx_encoder_wait_stream", "mx_encoder_destroy", "mx_encoder_encode",
    "mx_encoder_encode_device", "mx_encoder_set_profiling", "mx_encoder_get_stats",
    "mx_encoder_reset_stats",
    "mx_tokenizer_create", "mx_tokenizer_create_from_memory", "mx_tokenizer_create_bpe", "mx_tokenizer_create_bpe_from_memory",
    "mx_tokenizer_create_from_json", "mx_tokenizer_create_from_json_memory",
    "mx_tokenizer_destroy", "mx_tokenizer_vocab_size",
    "mx_tokenizer_encode", "mx_tokenizer_decode", "mx_tokenizer_segment", "mx_tokenizer_encode_batch",
    "mx_tokenizer_segment_batch", "mx_tokenizer_encode_staged",
]


class MemexHipError(RuntimeError):
    """A C-ABI call returned a negative mx_status."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"[mx_status {code}] {msg}")
        self.code = code
        self.msg = msg


class IndexStats(ctypes.Structure):
    _fields_ = [("searches", ctypes.c_uint64), ("queries", ctypes.c_uint64),
                ("fallback_queries", ctypes.c_uint64), ("scan_launches", ctypes.c_uint64),
                ("scan_bytes", ctypes.c_uint64), ("scan_ms", ctypes.c_double),
                ("candidates", ctypes.c_uint64), ("max_abs_err", ctypes.c_double),
                ("filter_copy_bytes", ctypes.c_uint64), ("retry_queries", ctypes.c_uint64),
                ("approx_err_bound", ctypes.c_double), ("filter_kind", ctypes.c_uint64),
                ("filter_demotions", ctypes.c_uint64), ("filter_promotions", ctypes.c_uint64),
                ("listed_rows", ctypes.c_uint64), ("exchange_fallbacks", ctypes.c_uint64),
                ("filter_centred", ctypes.c_uint64), ("exchange_ms", ctypes.c_double)]


class EncoderCfg(ctypes.Structure):
    _fields_ = [("layers", ctypes.c_int32), ("hidden", ctypes.c_int32), ("heads", ctypes.c_int32),
                ("ffn", ctypes.c_int32), ("vocab", ctypes.c_int32), ("max_pos", ctypes.c_int32),
                ("type_vocab", ctypes.c_int32), ("ln_eps", ctypes.c_float), ("pooling", ctypes.c_int32),
                ("normalize", ctypes.c_int32), ("pos_offset", ctypes.c_int32), ("precision", ctypes.c_int32)]


class EncoderStats(ctypes.Structure):
    _fields_ = [("calls", ctypes.c_uint64), ("sequences", ctypes.c_uint64), ("tokens", ctypes.c_uint64),
                ("flops", ctypes.c_double), ("gpu_ms", ctypes.c_double)]


def build(force: bool = False) -> str:
    """Compile libmemex_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-j8", "-s"]
    if force:
        subprocess.check_call(["make", "-C", CSRC, "-s", "clean"])
    subprocess.check_call(args)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("build did not produce " + LIB_PATH)
    return LIB_PATH


_lib = None
_lock = threading.Lock()


def _declare(L: ctypes.CDLL) -> None:
    vp, i32, u64, cp = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_char_p
    P = ctypes.POINTER
    L.mx_last_error.restype = cp
    L.mx_last_error.argtypes = []
    L.mx_version.restype = cp
    L.mx_version.argtypes = []
    sig = {
        "mx_device_count": [P(i32)],
        "mx_index_open": [cp, i32, i32, P(vp)],
        "mx_index_open_sharded": [cp, i32, i32, P(i32), u64, P(vp)],
        "mx_index_n_shards": [vp, P(i32)],
        "mx_index_exchange": [vp, P(i32)],
        "mx_index_wait_stream": [vp, vp],
        "mx_index_dim": [vp, P(i32)],
        "mx_index_size": [vp, P(u64)],
        "mx_index_reserve": [vp, u64],
        "mx_index_set_id_offset": [vp, u64],
        "mx_index_add": [vp, vp, u64, P(u64)],
        "mx_index_add_device": [vp, vp, u64, P(u64)],
        "mx_index_clear": [vp],
        "mx_index_search": [vp, vp, i32, i32, vp, vp, vp, vp],
        "mx_index_search_device": [vp, vp, i32, i32, vp, vp, vp, vp],
        "mx_index_set_search_mode": [vp, i32],
        "mx_index_set_filter_copy": [vp, i32],
        "mx_index_set_corpus_mode": [vp, i32],
        "mx_index_get_rows": [vp, u64, u64, vp],
        "mx_index_save": [vp, cp],
        "mx_index_load": [vp, cp],
        "mx_index_has_store": [cp, P(i32)],
        "mx_index_store_info": [cp, P(i32), P(u64)],
        "mx_index_remove_files": [cp],
        "mx_index_set_profiling": [vp, i32],
        "mx_index_get_stats": [vp, P(IndexStats)],
        "mx_index_reset_stats": [vp],
        "mx_topk_merge_device": [i32, vp, vp, i32, i32, i32, vp, vp, vp],
        "mx_topk_merge_packed_device": [i32, vp, i32, i32, i32, vp, vp, vp],
        "mx_topk_merge_packed_async": [i32, vp, vp, i32, i32, i32, vp, vp, vp],
        "mx_encoder_create": [P(EncoderCfg), vp, ctypes.c_size_t, i32, P(vp)],
        "mx_encoder_open": [cp, P(EncoderCfg), vp, ctypes.c_size_t, i32, P(vp)],
        "mx_encoder_wait_stream": [vp, vp],
        "mx_encoder_encode": [vp, vp, vp, i32, i32, vp],
        "mx_encoder_encode_device": [vp, vp, vp, i32, i32, vp],
        "mx_encoder_set_profiling": [vp, i32],
        "mx_encoder_get_stats": [vp, P(EncoderStats)],
        "mx_encoder_reset_stats": [vp],
        "mx_tokenizer_create": [cp, i32, P(vp)],
        "mx_tokenizer_create_from_memory": [cp, ctypes.c_size_t, i32, P(vp)],
        "mx_tokenizer_create_bpe": [cp, cp, P(vp)],
        "mx_tokenizer_create_bpe_from_memory": [cp, ctypes.c_size_t, cp, ctypes.c_size_t, P(vp)],
        "mx_tokenizer_create_from_json": [cp, P(vp)],
        "mx_tokenizer_create_from_json_memory": [cp, ctypes.c_size_t, P(vp)],
        "mx_tokenizer_vocab_size": [vp, P(i32)],
        "mx_tokenizer_encode": [vp, cp, i32, vp, i32, P(i32)],
        "mx_tokenizer_decode": [vp, vp, i32, i32, vp, ctypes.c_size_t, P(ctypes.c_size_t)],
        "mx_tokenizer_segment": [vp, cp, i32, i32, vp, ctypes.c_size_t, P(ctypes.c_size_t), P(i32)],
        "mx_tokenizer_encode_batch": [vp, P(cp), i32, i32, vp, i32, vp, P(i32)],
        "mx_tokenizer_segment_batch": [vp, P(cp), i32, i32, i32, vp, ctypes.c_size_t, P(ctypes.c_size_t), vp],
        "mx_tokenizer_encode_staged": [vp, cp, vp, i32, P(i32)],
    }
    for name, argtypes in sig.items():
        fn = getattr(L, name)
        fn.restype = i32
        fn.argtypes = argtypes
    L.mx_index_close.restype = None
    L.mx_index_close.argtypes = [vp]
    L.mx_encoder_destroy.restype = None
    L.mx_encoder_destroy.argtypes = [vp]
    L.mx_tokenizer_destroy.restype = None
    L.mx_tokenizer_destroy.argtypes = [vp]
    L.mx_encoder_weight_bytes.restype = ctypes.c_size_t
    L.mx_encoder_weight_bytes.argtypes = [P(EncoderCfg)]
    L.mx_index_stats_size.restype = ctypes.c_size_t
    L.mx_index_stats_size.argtypes = []
    L.mx_encoder_cfg_size.restype = ctypes.c_size_t
    L.mx_encoder_cfg_size.argtypes = []


def _load_torch_runtime_first() -> None:
    """PyTorch wheels bundle their own libamdhip64 / libhsa-runtime64 (ROCm 7.0) while
    libmemex_hip.so links the system ROCm runtime.  Both coexist in one process (device memory is
    process-wide), but only when torch's copies are loaded FIRST: with the opposite order torch's
    lazy CUDA init later reports "No HIP GPUs are available".  So, in processes where torch is
    installed, import it before dlopen()ing our library.  Processes without torch (the Rust host)
    are unaffected."""
    import importlib.util
    import sys
    if "torch" not in sys.modules and importlib.util.find_spec("torch") is not None:
        try:
            import torch  # noqa: F401
        except Exception:
            pass


_path = LIB_PATH


def use_testing_library() -> None:
    """Tests only, before the first lib() of the process: bind the package to libmemex_hip_testing.so."""
    global _path
    with _lock:
        if _lib is not None:
            raise RuntimeError("the library is already loaded")
        _path = TESTING_LIB_PATH


def testing_lib() -> ctypes.CDLL:
    """Tests only: libmemex_hip_testing.so as a second, separately loaded library (own globals), declared like lib()."""
    _load_torch_runtime_first()
    L = ctypes.CDLL(TESTING_LIB_PATH)
    _declare(L)
    return L


def lib() -> ctypes.CDLL:
    """The loaded library.  Raises (never falls back) when it is absent."""
    global _lib
    with _lock:
        if _lib is None:
            _load_torch_runtime_first()
            if not os.path.exists(_path):
                raise MemexHipError(MX_EDEVICE, f"{_path} is missing: run memex_amd.build() "
                                                "(python -c 'import __graft_entry__ as g; g.build()')")
            L = ctypes.CDLL(_path)
            _declare(L)
            # ABI handshake: mx_index_stats grows at the end from release to release; a binding older or newer than the
            # library must not read past / short of what the library writes
            if L.mx_index_stats_size() != ctypes.sizeof(IndexStats):
                raise MemexHipError(MX_EINVAL, f"{LIB_PATH}: mx_index_stats is {L.mx_index_stats_size()} bytes, this binding "
                                               f"declares {ctypes.sizeof(IndexStats)} (library / binding version mismatch)")
            if L.mx_encoder_cfg_size() != ctypes.sizeof(EncoderCfg):
                raise MemexHipError(MX_EINVAL, f"{_path}: mx_encoder_cfg is {L.mx_encoder_cfg_size()} bytes, this binding "
                                               f"declares {ctypes.sizeof(EncoderCfg)} (library / binding version mismatch)")
            _lib = L
        return _lib


def check(rc: int) -> None:
    if rc != MX_OK:
        raise MemexHipError(rc, lib().mx_last_error().decode("utf-8", "replace"))


def device_count() -> int:
    n = ctypes.c_int(0)
    rc = lib().mx_device_count(ctypes.byref(n))
    return n.value if rc == MX_OK else 0
