"""Encoder configurations and weight packing for ``mx_encoder_create``.

The blob layout (f32, HF ``BertModel`` tensor names, Linear weights ``[out, in]``) is documented in
``include/memex_hip.h``.  ``pack_weights`` accepts any mapping name -> array (a torch ``state_dict``
with or without the ``bert.`` prefix, a safetensors file via :func:`load_safetensors`);
``synthetic_weights`` produces seeded weights of the real shapes -- pretrained checkpoints are not
reachable offline, so parity work and benchmarks run on these.

Reference: the model zoo is selected by ``EmbeddingsModelType`` / ``ModelConfig``
(lib/libmemex/src/llm/embedding.rs:25-73); the default is all-MiniLM-L12-v2.
"""
from __future__ import annotations

from dataclasses import dataclass, asdict
from typing import Dict, List, Mapping, Tuple

import numpy as np


@dataclass(frozen=True)
class EncoderConfig:
    layers: int
    hidden: int
    heads: int
    ffn: int
    vocab: int = 30522
    max_pos: int = 512
    type_vocab: int = 2
    ln_eps: float = 1e-12
    pooling: str = "mean"      # "mean" (MiniLM) | "cls" (bge)
    normalize: bool = True
    max_seq_length: int = 256  # sentence_bert_config.json truncation
    pos_offset: int = 0        # 0 = BERT; 2 = RoBERTa-style (position ids start at padding_idx + 1)
    precision: str = "bf16"    # "bf16" (default: bf16 operands, the ingest path) | "bf16x3" (split operands, f32 hidden state
                               # and attention: f32-grade scores at ~1/3 of the speed; include/memex_hip.h MX_PREC_BF16X3) |
                               # "mixed" (MX_PREC_MIXED: bf16x3 with the MLP on two fp16 products: 15-17 % faster, scores ~1e-3 at worst) |
                               # "mixed1" (MX_PREC_MIXED1: ... on ONE fp16 product: another ~20 % faster, scores 1e-5 ... 2e-2 by checkpoint; opt-in)

    def as_dict(self) -> dict:
        return asdict(self)


ALL_MINILM_L6_V2 = EncoderConfig(layers=6, hidden=384, heads=12, ffn=1536, max_seq_length=256)
ALL_MINILM_L12_V2 = EncoderConfig(layers=12, hidden=384, heads=12, ffn=1536, max_seq_length=128)
BGE_BASE_EN = EncoderConfig(layers=12, hidden=768, heads=12, ffn=3072, pooling="cls", max_seq_length=512)
# all-distilroberta-v1 (embedding.rs:29): DistilRoBERTa = 6 RoBERTa layers.  Same encoder stack as
# BERT; the embeddings differ: one token type, 514 positions used from row 2 on, LayerNorm eps 1e-5.
# Its tokenizer is byte-level BPE (vocab.json + merges.txt): memex_amd.tokenizer.ByteLevelBpeTokenizer; segment_text accepts
# the model (embedding.rs:159).
ALL_DISTILROBERTA_V1 = EncoderConfig(layers=6, hidden=768, heads=12, ffn=3072, vocab=50265, max_pos=514, type_vocab=1,
                                     ln_eps=1e-5, max_seq_length=512, pos_offset=2)


def tensor_order(cfg: EncoderConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    H, F = cfg.hidden, cfg.ffn
    out = [("embeddings.word_embeddings.weight", (cfg.vocab, H)),
           ("embeddings.position_embeddings.weight", (cfg.max_pos, H)),
           ("embeddings.token_type_embeddings.weight", (cfg.type_vocab, H)),
           ("embeddings.LayerNorm.weight", (H,)), ("embeddings.LayerNorm.bias", (H,))]
    for l in range(cfg.layers):
        p = f"encoder.layer.{l}."
        for n in ("query", "key", "value"):
            out += [(p + f"attention.self.{n}.weight", (H, H)), (p + f"attention.self.{n}.bias", (H,))]
        out += [(p + "attention.output.dense.weight", (H, H)), (p + "attention.output.dense.bias", (H,)),
                (p + "attention.output.LayerNorm.weight", (H,)), (p + "attention.output.LayerNorm.bias", (H,)),
                (p + "intermediate.dense.weight", (F, H)), (p + "intermediate.dense.bias", (F,)),
                (p + "output.dense.weight", (H, F)), (p + "output.dense.bias", (H,)),
                (p + "output.LayerNorm.weight", (H,)), (p + "output.LayerNorm.bias", (H,))]
    return out


def pack_weights(state: Mapping[str, object], cfg: EncoderConfig) -> np.ndarray:
    """-> contiguous f32 blob in the order ``mx_encoder_create`` expects."""
    parts = []
    for name, shape in tensor_order(cfg):
        t = None
        for key in (name, "bert." + name, "roberta." + name, "0.auto_model." + name, "model." + name):
            if key in state:
                t = state[key]
                break
        if t is None:
            raise KeyError(f"weight '{name}' missing")
        a = t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)
        a = np.asarray(a, dtype=np.float32)
        if tuple(a.shape) != shape:
            raise ValueError(f"{name}: shape {a.shape}, expected {shape}")
        parts.append(a.reshape(-1))
    return np.ascontiguousarray(np.concatenate(parts))


def synthetic_weights(cfg: EncoderConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    """Seeded weights with the real shapes (BERT-style init scales; LayerNorm near identity)."""
    rng = np.random.default_rng(seed)
    out: Dict[str, np.ndarray] = {}
    for name, shape in tensor_order(cfg):
        if name.endswith("LayerNorm.weight"):
            a = 1.0 + 0.1 * rng.standard_normal(shape)
        elif name.endswith("LayerNorm.bias"):
            a = 0.05 * rng.standard_normal(shape)
        elif name.endswith(".bias"):
            a = 0.02 * rng.standard_normal(shape)
        elif "embeddings" in name:
            a = 0.05 * rng.standard_normal(shape)
        else:
            a = rng.standard_normal(shape) * (1.0 / np.sqrt(shape[1]))  # keeps activations O(1)
        out[name] = a.astype(np.float32)
    return out


def checkpoint_like_weights(cfg: EncoderConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    """``synthetic_weights`` with the features of TRAINED BERT-family checkpoints that are the known hazards of a bf16
    forward pass (real weights are unreachable offline: the reference downloads them, embedding.rs:99-100): a few
    "outlier" hidden dimensions carried through every layer -- LayerNorm gains of ~20 and biases of +-30 on the same
    handful of dimensions in every LayerNorm, as in BERT / RoBERTa checkpoints -- and query / key projections scaled so
    that attention logits reach +-60.  Used by the stress parity tests."""
    w = synthetic_weights(cfg, seed)
    rng = np.random.default_rng(seed + 1000)
    dims = rng.choice(cfg.hidden, size=5, replace=False)
    for name in list(w):
        if name.endswith("LayerNorm.weight"):
            w[name][dims[:3]] *= np.float32(20.0)
        elif name.endswith("LayerNorm.bias"):
            w[name][dims[3]] = np.float32(30.0)
            w[name][dims[4]] = np.float32(-30.0)
        elif name.endswith("attention.self.query.weight") or name.endswith("attention.self.key.weight"):
            w[name] = (w[name] * np.float32(4.0)).astype(np.float32)
    return w


def load_safetensors(path: str) -> Dict[str, np.ndarray]:
    from safetensors.numpy import load_file
    return load_file(path)
