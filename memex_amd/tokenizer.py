"""Native WordPiece tokenizer + sliding-window segmenter (``mx_tokenizer_*``, host code in
``csrc/tokenizer.cpp``): what memex does with the ``tokenizers`` crate in ``segment_text``
(reference lib/libmemex/src/llm/embedding.rs:155-198) and what rust-bert does before the forward."""
from __future__ import annotations

import ctypes
from typing import List, Sequence, Tuple

import numpy as np

from ._lib import check, lib


class _NativeTokenizer:
    """What both kinds of ``mx_tokenizer`` handle offer (the calls of segment_text and of rust-bert's tokenisation)."""
    _h = None

    def _finish(self, h) -> None:
        self._h = h
        n = ctypes.c_int(0)
        check(lib().mx_tokenizer_vocab_size(self._h, ctypes.byref(n)))
        self.vocab = n.value

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().mx_tokenizer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def encode(self, text: str, add_special_tokens: bool = False) -> List[int]:
        raw = text.encode("utf-8").replace(b"\0", b"")
        cap = max(16, len(raw) + 2)
        ids = (ctypes.c_int32 * cap)()
        n = ctypes.c_int(0)
        check(lib().mx_tokenizer_encode(self._h, raw, 1 if add_special_tokens else 0, ids, cap, ctypes.byref(n)))
        return list(ids[: n.value])

    def decode(self, ids: Sequence[int], skip_special_tokens: bool = True) -> str:
        arr = (ctypes.c_int32 * max(1, len(ids)))(*ids)
        nb = ctypes.c_size_t(0)
        check(lib().mx_tokenizer_decode(self._h, arr, len(ids), 1 if skip_special_tokens else 0, None, 0, ctypes.byref(nb)))
        buf = ctypes.create_string_buffer(nb.value)
        check(lib().mx_tokenizer_decode(self._h, arr, len(ids), 1 if skip_special_tokens else 0, buf, nb.value, ctypes.byref(nb)))
        return buf.value.decode("utf-8")

    def windows(self, text: str, max_length: int, stride: int) -> List[str]:
        """segment_text: detokenised windows of ``max_length`` tokens overlapping by ``stride``."""
        return self.windows_batch([text], max_length, stride)[0]

    def windows_batch(self, texts: Sequence[str], max_length: int, stride: int) -> List[List[str]]:
        """segment_text for several documents in one call (``mx_tokenizer_segment_batch``: one host thread per document)."""
        n = len(texts)
        if n == 0:
            return []
        raws = [t.encode("utf-8").replace(b"\0", b"") for t in texts]
        arr = (ctypes.c_char_p * n)(*raws)
        nseg = (ctypes.c_int32 * n)()
        nb = ctypes.c_size_t(0)
        # decoded windows overlap by stride / (max_length - stride) and gain a space per isolated punctuation mark: this
        # estimate covers ordinary text in one call; the call reports the size it needs when it does not
        cap = int(sum(len(r) for r in raws) * (2.0 + stride / max(1, max_length - stride))) + 64 * n + 64
        for _ in range(2):
            buf = ctypes.create_string_buffer(cap)
            check(lib().mx_tokenizer_segment_batch(self._h, arr, n, max_length, stride, buf, cap, ctypes.byref(nb), nseg))
            if nb.value <= cap:
                break
            cap = nb.value
        parts = buf.raw[: nb.value].split(b"\0")
        out, o = [], 0
        for i in range(n):
            out.append([p.decode("utf-8") for p in parts[o:o + nseg[i]]])
            o += nseg[i]
        return out

    def encode_batch(self, texts: Sequence[str], max_seq_length: int) -> Tuple[np.ndarray, np.ndarray]:
        """-> (ids int32 [B,S], lens int32 [B]) with [CLS]/[SEP], truncated, [PAD]-padded to the batch max."""
        B = len(texts)
        arr = (ctypes.c_char_p * B)(*[t.encode("utf-8").replace(b"\0", b"") for t in texts])
        ids = np.zeros((B, max_seq_length), dtype=np.int32)
        lens = np.zeros(B, dtype=np.int32)
        S = ctypes.c_int(0)
        check(lib().mx_tokenizer_encode_batch(self._h, arr, B, max_seq_length, ids.ctypes.data_as(ctypes.c_void_p),
                                              max_seq_length, lens.ctypes.data_as(ctypes.c_void_p), ctypes.byref(S)))
        return np.ascontiguousarray(ids[:, : S.value]), lens


class WordPieceTokenizer(_NativeTokenizer):
    def __init__(self, vocab, lowercase: bool = True):
        """``vocab``: path to a BERT ``vocab.txt`` or a list of tokens (index = id)."""
        h = ctypes.c_void_p()
        if isinstance(vocab, str):
            check(lib().mx_tokenizer_create(vocab.encode(), 1 if lowercase else 0, ctypes.byref(h)))
        else:
            blob = "\n".join(vocab).encode("utf-8")
            check(lib().mx_tokenizer_create_from_memory(blob, len(blob), 1 if lowercase else 0, ctypes.byref(h)))
        self._finish(h)

    def encode_staged(self, text: str) -> List[int]:
        """The stage-by-stage encoder (test hook: ``encode`` runs the same steps in one pass)."""
        raw = text.encode("utf-8").replace(b"\0", b"")
        cap = max(16, len(raw) + 2)
        ids = (ctypes.c_int32 * cap)()
        n = ctypes.c_int(0)
        check(lib().mx_tokenizer_encode_staged(self._h, raw, ids, cap, ctypes.byref(n)))
        return list(ids[: n.value])


class ByteLevelBpeTokenizer(_NativeTokenizer):
    """The RoBERTa-family tokenizer (all-distilroberta-v1, embedding.rs:29,159): ``vocab.json`` + ``merges.txt``."""

    def __init__(self, vocab_json: str, merges: str):
        h = ctypes.c_void_p()
        check(lib().mx_tokenizer_create_bpe(vocab_json.encode(), merges.encode(), ctypes.byref(h)))
        self._finish(h)


class JsonTokenizer(_NativeTokenizer):
    """A native tokenizer built from ``tokenizer.json`` -- the file ``Tokenizer::from_pretrained`` itself reads
    (embedding.rs:163): WordPiece or byte-level BPE, whichever the file's ``model`` describes.  A component the native
    code does not implement raises :class:`memex_amd._lib.MemexHipError` with ``MX_EUNSUPPORTED``."""

    def __init__(self, tokenizer_json):
        """``tokenizer_json``: a path, or the document itself as ``bytes``."""
        h = ctypes.c_void_p()
        if isinstance(tokenizer_json, (bytes, bytearray)):
            blob = bytes(tokenizer_json)
            check(lib().mx_tokenizer_create_from_json_memory(blob, len(blob), ctypes.byref(h)))
        else:
            check(lib().mx_tokenizer_create_from_json(str(tokenizer_json).encode(), ctypes.byref(h)))
        self._finish(h)
