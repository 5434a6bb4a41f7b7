"""Host-side mirror of memex's sentence-embedder actor, running on the HIP encoder.

Same names and behaviour as the reference (lib/libmemex/src/llm/embedding.rs):

=========================  ==================================================================
reference                  here
=========================  ==================================================================
``EmbeddingError``         :class:`EmbeddingError`, :class:`EncodingFailure`, :class:`SetupError` (:11-16)
``EmbeddingResult``        :class:`EmbeddingResult`                                  (:19-22)
``EmbeddingsModelType``    :class:`EmbeddingsModelType`                              (:25-33)
``ModelConfig``            :class:`ModelConfig` (default L12-v2, 256, 86)            (:58-73)
``SentenceEmbedder``       :class:`SentenceEmbedder` -- ``spawn`` / ``encode`` / ``encode_single`` (:78-152):
                           a dedicated thread owns the model, fed through a bounded queue of 100
``segment_text``           :func:`segment_text` -- 256-token windows, 86-token overlap (:155-198)
=========================  ==================================================================

Tokenisation stays on the host.  The reference downloads the pretrained ``tokenizer.json`` from the
HF hub (embedding.rs:163); that is impossible offline, so the tokenizer is passed in: a path to a BERT
``vocab.txt`` selects the NATIVE WordPiece tokenizer / segmenter (``mx_tokenizer_*``,
``csrc/tokenizer.cpp``, parity-tested against the ``tokenizers`` package), a pair ``(vocab.json, merges.txt)`` the native
byte-level BPE tokenizer (all-distilroberta-v1), a ``tokenizer.json`` path or
``tokenizers.Tokenizer`` object is adapted, and with nothing given :class:`WhitespaceHashTokenizer` --
a clearly labelled STAND-IN with the same windowing arithmetic -- keeps the plumbing runnable.
"""
from __future__ import annotations

import enum
import queue
import threading
import zlib
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import weights as W
from .encoder import Encoder


class EmbeddingError(Exception):
    pass


class EncodingFailure(EmbeddingError):
    pass


class SetupError(EmbeddingError):
    pass


@dataclass
class EmbeddingResult:
    """embedding.rs:18-22.  ``vector``: the reference's ``Vec<f32>`` -- here one float32 row (``numpy.ndarray``, indexable and
    iterable like a list, its own copy -- not a view of the batch; a Python list of 384 floats per window was a third of a
    document's ingest time).  NOT a list where it matters: `if r.vector` raises, `==` is elementwise, `json.dumps` needs
    `r.vector.tolist()` (`VectorData.to_json()` does that at the serialisation boundary)."""
    content: str
    vector: Sequence[float]


class EmbeddingsModelType(enum.Enum):
    DistiluseBaseMultilingualCased = "distiluse-base-multilingual-cased"
    BertBaseNliMeanTokens = "bert-base-nli-mean-tokens"
    AllMiniLmL12V2 = "all-MiniLM-L12-v2"
    AllMiniLmL6V2 = "all-MiniLM-L6-v2"
    AllDistilrobertaV1 = "all-distilroberta-v1"
    ParaphraseAlbertSmallV2 = "paraphrase-albert-small-v2"
    SentenceT5Base = "sentence-t5-base"


_ENCODER_CONFIGS = {
    EmbeddingsModelType.AllMiniLmL12V2: W.ALL_MINILM_L12_V2,
    EmbeddingsModelType.AllMiniLmL6V2: W.ALL_MINILM_L6_V2,
    EmbeddingsModelType.AllDistilrobertaV1: W.ALL_DISTILROBERTA_V1,
}


@dataclass(frozen=True)
class ModelConfig:
    model: EmbeddingsModelType = EmbeddingsModelType.AllMiniLmL12V2  # embedding.rs:67
    max_length: int = 256                                             # :68
    stride: int = 86                                                  # :70 "roughly a third"


class WhitespaceHashTokenizer:
    """STAND-IN for the pretrained WordPiece tokenizer (not obtainable offline): lower-cases, splits
    on whitespace, hashes each word into [1000, vocab).  Same special ids as BERT ([PAD]=0,
    [CLS]=101, [SEP]=102) and the same truncation/stride window arithmetic as tokenizers 0.14."""
    pad_id, cls_id, sep_id = 0, 101, 102

    def __init__(self, vocab: int = 30522):
        self.vocab = vocab

    def words(self, text: str) -> List[str]:
        return text.lower().split()

    def word_id(self, w: str) -> int:
        return 1000 + zlib.crc32(w.encode("utf-8")) % (self.vocab - 1000)

    def windows(self, text: str, max_length: int, stride: int) -> List[str]:
        ws = self.words(text)
        if not ws:
            return [""]
        step = max_length - stride          # tokenizers: each overflow window starts max_length-stride later
        out, start = [], 0
        while True:
            out.append(" ".join(ws[start:start + max_length]))
            if start + max_length >= len(ws):
                break
            start += step
        return out

    def encode_batch(self, texts: Sequence[str], max_seq_length: int) -> Tuple[np.ndarray, np.ndarray]:
        rows = []
        for t in texts:
            ids = [self.word_id(w) for w in self.words(t)][: max_seq_length - 2]
            rows.append([self.cls_id] + ids + [self.sep_id])
        S = max(len(r) for r in rows)
        ids = np.full((len(rows), S), self.pad_id, dtype=np.int32)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = r
        return ids, np.asarray([len(r) for r in rows], dtype=np.int32)


class HFTokenizerAdapter:
    """``tokenizers.Tokenizer`` (e.g. loaded from a local tokenizer.json) with the reference's calls."""

    def __init__(self, tok):
        self.tok = tok

    def windows(self, text: str, max_length: int, stride: int) -> List[str]:
        self.tok.enable_truncation(max_length=max_length, stride=stride)        # embedding.rs:173-177
        self.tok.no_padding()
        enc = self.tok.encode(text, add_special_tokens=False)                   # :181
        out = [self.tok.decode(enc.ids, skip_special_tokens=True).replace(" ' ", "'")]  # :182-183
        for o in enc.overflowing:                                               # :188-195 (no replace)
            out.append(self.tok.decode(o.ids, skip_special_tokens=True))
        return out

    def encode_batch(self, texts: Sequence[str], max_seq_length: int):
        self.tok.enable_truncation(max_length=max_seq_length)
        self.tok.no_padding()
        encs = self.tok.encode_batch(list(texts), add_special_tokens=True)
        S = max(len(e.ids) for e in encs)
        ids = np.zeros((len(encs), S), dtype=np.int32)
        for i, e in enumerate(encs):
            ids[i, :len(e.ids)] = e.ids
        return ids, np.asarray([len(e.ids) for e in encs], dtype=np.int32)


def _as_tokenizer(tokenizer, vocab: int):
    if tokenizer is None:
        return WhitespaceHashTokenizer(vocab)
    if isinstance(tokenizer, (tuple, list)) and len(tokenizer) == 2:        # (vocab.json, merges.txt): native byte-level BPE
        from .tokenizer import ByteLevelBpeTokenizer
        try:
            return ByteLevelBpeTokenizer(*tokenizer)
        except Exception as e:
            raise SetupError(f"Unable to load model <{tokenizer[0]}>") from e
    if isinstance(tokenizer, str) and tokenizer.endswith(".txt"):
        from .tokenizer import WordPieceTokenizer           # native WordPiece over a BERT vocab.txt
        try:
            return WordPieceTokenizer(tokenizer, lowercase=True)
        except Exception as e:
            raise SetupError(f"Unable to load model <{tokenizer}>") from e
    if isinstance(tokenizer, str):                           # tokenizer.json: native when it is one of the two stacks the
        from ._lib import MX_EUNSUPPORTED, MemexHipError    # library implements, the `tokenizers` package for anything else
        from .tokenizer import JsonTokenizer
        try:
            return JsonTokenizer(tokenizer)
        except MemexHipError as e:
            if e.code != MX_EUNSUPPORTED:
                raise SetupError(f"Unable to load model <{tokenizer}>") from e
        try:
            from tokenizers import Tokenizer
            return HFTokenizerAdapter(Tokenizer.from_file(tokenizer))
        except Exception as e:
            raise SetupError(f"Unable to load model <{tokenizer}>") from e
    if hasattr(tokenizer, "windows") and hasattr(tokenizer, "encode_batch"):
        return tokenizer
    return HFTokenizerAdapter(tokenizer)


_SEGMENTABLE = (EmbeddingsModelType.AllMiniLmL12V2, EmbeddingsModelType.AllMiniLmL6V2,
                EmbeddingsModelType.AllDistilrobertaV1)                          # embedding.rs:156-161


def segment_text(model_config: ModelConfig, text: str, tokenizer=None) -> List[str]:
    """embedding.rs:155-198: sliding windows of ``max_length`` tokens overlapping by ``stride``."""
    if model_config.model not in _SEGMENTABLE:
        raise SetupError("Model not supported yet")                              # :160
    tok = _as_tokenizer(tokenizer, 30522)
    try:
        return tok.windows(text, model_config.max_length, model_config.stride)
    except EmbeddingError:
        raise
    except Exception as e:
        raise EncodingFailure(text) from e


class SentenceEmbedder:
    """embedding.rs:78-152.  ``spawn`` starts the dedicated model thread and returns
    ``(thread, embedder)``; messages are ``(text, segment?, reply)`` on a queue bounded at 100."""

    def __init__(self, q: "queue.Queue"):
        self._q = q

    @classmethod
    def spawn(cls, model_config: ModelConfig = ModelConfig(), weights=None, tokenizer=None, device: int = 0,
              encoder_config: Optional[W.EncoderConfig] = None, seed: int = 0, allow_synthetic: bool = False,
              encoder_key: Optional[str] = None):
        """``weights``: HF tensor mapping / packed blob of the checkpoint; ``tokenizer``: path to its
        ``vocab.txt`` (native WordPiece) or ``tokenizer.json``, or a tokenizer object.  Like the
        reference (``create_model()`` / ``Tokenizer::from_pretrained`` failing -> SetupError,
        embedding.rs:99-100,163-169) a missing checkpoint or vocabulary is an error; the seeded
        stand-ins (synthetic weights, whitespace-hash tokenizer) used by tests and benchmarks must be
        asked for with ``allow_synthetic=True``."""
        if (weights is None or tokenizer is None) and not allow_synthetic:
            what = "weights" if weights is None else "tokenizer"
            raise SetupError(f"Unable to load model <{model_config.model.value}>: no {what} given "
                             "(pass allow_synthetic=True for the seeded stand-ins)")
        q: "queue.Queue" = queue.Queue(maxsize=100)                              # sync_channel(100), :87
        ready: "queue.Queue" = queue.Queue(maxsize=1)
        th = threading.Thread(target=cls._runner, args=(q, ready, model_config, weights, tokenizer, device,
                                                        encoder_config, seed, encoder_key), daemon=True)
        th.start()
        err = ready.get()
        if err is not None:
            raise err
        return th, cls(q)

    @classmethod
    def from_pretrained_dir(cls, path: str, model_config: ModelConfig = ModelConfig(), device: int = 0,
                            precision: Optional[str] = None, encoder_key: Optional[str] = None):
        """``create_model()`` from a LOCAL sentence-transformers directory -- the files
        ``SentenceEmbeddingsBuilder::remote(..)`` downloads (embedding.rs:99-100): ``modules.json``, ``config.json``,
        ``sentence_bert_config.json``, ``1_Pooling/config.json``, ``model.safetensors`` / ``pytorch_model.bin``,
        ``vocab.txt`` (:mod:`memex_amd.pretrained`).  Returns ``(thread, embedder)`` like :meth:`spawn`; a model the HIP
        encoder does not run (a ``2_Dense`` module, max pooling, ...) raises :class:`SetupError`."""
        from .pretrained import UnsupportedModel, load_pretrained_dir
        try:
            cfg, tensors, vocab, info = load_pretrained_dir(path, precision)
        except (UnsupportedModel, OSError, KeyError, ValueError) as e:
            raise SetupError(f"Unable to load model <{path}>: {e}") from e
        if vocab is None and info["bpe_files"] is None and info.get("tokenizer_json") is None:
            raise SetupError(f"Unable to load model <{path}>: neither vocab.txt (WordPiece), vocab.json + merges.txt "
                             "(byte-level BPE) nor tokenizer.json")
        from .tokenizer import ByteLevelBpeTokenizer, JsonTokenizer, WordPieceTokenizer
        try:
            tok = (WordPieceTokenizer(vocab, lowercase=info["do_lower_case"]) if vocab is not None
                   else ByteLevelBpeTokenizer(*info["bpe_files"]) if info["bpe_files"] is not None
                   else JsonTokenizer(info["tokenizer_json"]))
        except Exception as e:
            raise SetupError(f"Unable to load model <{path}>: {e}") from e
        return cls.spawn(model_config, weights=tensors, tokenizer=tok, device=device, encoder_config=cfg,
                         encoder_key=encoder_key)

    @staticmethod
    def _runner(q, ready, model_config, weights, tokenizer, device, encoder_config, seed, encoder_key=None):
        try:
            cfg = encoder_config or _ENCODER_CONFIGS.get(model_config.model)
            if cfg is None:
                raise SetupError("Model not supported yet")
            if weights is None:  # no pretrained checkpoint offline: seeded synthetic weights of the real shape
                weights = W.synthetic_weights(cfg, seed)
            tok = _as_tokenizer(tokenizer, cfg.vocab)
            # create_model(), :99-100 -- keyed: embedders spawned per request / per task (handlers.rs:61-63,
            # tasks.rs:17) share one resident copy of the weights instead of re-uploading them
            enc = Encoder(cfg, weights, device, key=encoder_key)
        except Exception as e:  # surfaces like RustBertError from runner
            ready.put(e if isinstance(e, EmbeddingError) else SetupError(str(e)))
            return
        ready.put(None)
        # Two stages, one thread each, a slot of one batch between them: while the GPU embeds batch i (a C call, no GIL) this
        # thread segments and tokenises batch i+1 -- host work per document (segment + tokenise its windows) is about as long
        # as its encoder pass, so the stages hide each other.  The reference's runner does both in turn per message (:101-109).
        staged: "queue.Queue" = queue.Queue(maxsize=1)

        def gpu_stage():
            while True:
                item = staged.get()
                if item is None:
                    return
                work, flat, ids, lens = item
                busy.set()
                try:
                    vecs = enc.encode(ids, lens)                                 # model.encode(&segments), :109
                    if len(vecs) != len(flat):
                        raise EncodingFailure("# of embeddings doesn't match # of segments")
                    o = 0
                    for reply, segs in work:
                        # (a copy per row: a view would keep the whole batch's [N, H] array alive behind every result)
                        reply.put([EmbeddingResult(content=s_, vector=v.copy())
                                   for s_, v in zip(segs, vecs[o:o + len(segs)])])
                        o += len(segs)
                except Exception as e:
                    for reply, _ in work:
                        reply.put(e)
                finally:
                    busy.clear()
        busy = threading.Event()
        gpu = threading.Thread(target=gpu_stage, daemon=True)
        gpu.start()
        stop = False
        while not stop:
            msgs = [q.get()]
            # Requests that queued up while the previous batch was on the GPU are embedded together
            # (SURVEY section 8 f-2): one tokenizer call, one encoder call.  The reference's runner takes
            # one message per model.encode (:101-109); a row's embedding does not depend on its batch.
            # With the GPU stage idle only half of what is waiting is taken: synchronous callers (the worker's five tasks) come
            # back with their next document only after a reply, so two smaller batches keep both stages busy where one batch of
            # everything would leave this stage waiting for the GPU and then the GPU waiting for this stage.
            limit = 64 if (busy.is_set() or not staged.empty()) else max(1, (1 + q.qsize()) // 2)
            while len(msgs) < limit:
                try:
                    msgs.append(q.get_nowait())
                except queue.Empty:
                    break
            work = []                                                            # (reply, segments)
            live = [m for m in msgs if m is not None]
            stop = len(live) != len(msgs)
            # the documents of the drained requests are segmented together (mx_tokenizer_segment_batch: one host thread per
            # document) -- at the encoder's rate the segmenter is the ingest bottleneck
            presegmented = {}
            to_seg = [i for i, (_, segment, _) in enumerate(live) if segment]
            if len(to_seg) > 1 and hasattr(tok, "windows_batch") and model_config.model in _SEGMENTABLE:
                try:
                    for i, segs in zip(to_seg, tok.windows_batch([live[i][0] for i in to_seg], model_config.max_length,
                                                                 model_config.stride)):
                        presegmented[i] = segs
                except Exception:
                    presegmented = {}                                            # one bad text: fall back to per-request errors
            for i, (text, segment, reply) in enumerate(live):
                try:
                    segs = presegmented[i] if i in presegmented else \
                        segment_text(model_config, text, tok) if segment else [text]      # :103-107
                    work.append((reply, segs))
                except Exception as e:
                    reply.put(e)
            if not work:
                continue
            try:
                flat = [s_ for _, segs in work for s_ in segs]
                ids, lens = tok.encode_batch(flat, cfg.max_seq_length)
            except Exception as e:
                for reply, _ in work:
                    reply.put(e)
                continue
            staged.put((work, flat, ids, lens))
        staged.put(None)
        gpu.join()
        enc.close()

    def _call(self, text: str, segment: bool):
        reply: "queue.Queue" = queue.Queue(maxsize=1)                            # oneshot channel, :139
        self._q.put((text, segment, reply))
        res = reply.get()
        if isinstance(res, Exception):
            raise res
        return res

    def encode(self, text: str) -> List[EmbeddingResult]:
        return self._call(text, True)                                            # :138-142

    def encode_single(self, text: str) -> Optional[EmbeddingResult]:
        res = self._call(text, False)                                            # :146-151
        return res.pop() if res else None

    def shutdown(self) -> None:
        self._q.put(None)
