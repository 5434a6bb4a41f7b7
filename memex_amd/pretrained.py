"""Load what the reference loads: a sentence-transformers model directory.

``SentenceEmbeddingsBuilder::remote(model_type).create_model()`` (reference
lib/libmemex/src/llm/embedding.rs:99-100; rust-bert 0.21.0) fetches, per model,

    modules.json                      the pipeline: Transformer -> Pooling [-> Dense] [-> Normalize]
    config.json                       the transformer's (BERT / RoBERTa) architecture
    sentence_bert_config.json         max_seq_length, do_lower_case
    1_Pooling/config.json             pooling mode
    model.safetensors | pytorch_model.bin | rust_model.ot      weights
    vocab.txt (+ tokenizer_config.json)                        WordPiece vocabulary   (BERT family)
    vocab.json + merges.txt                                    byte-level BPE         (RoBERTa family: all-distilroberta-v1)
    tokenizer.json                                             either kind in one file (what ``Tokenizer::from_pretrained``
                                                               reads in segment_text, embedding.rs:163); used when the
                                                               files above are absent

and builds the model from them.  :func:`load_pretrained_dir` reads the same files from a LOCAL directory (this build has no
network: the day a checkpoint is reachable the path is pointed at its directory) into an :class:`EncoderConfig`, the tensor
mapping ``pack_weights`` takes, and the vocabulary path for the native tokenizer.  What the HIP encoder does not implement is
refused with :class:`UnsupportedModel` (the Python face of ``MX_EUNSUPPORTED``), never approximated: a ``2_Dense`` module,
max / sqrt-length pooling, non-GELU activations, relative position embeddings, ``rust_model.ot`` archives (a libtorch
pickle: convert with ``safetensors`` first).
"""
from __future__ import annotations

import json
import os
from typing import Dict, Optional, Tuple

import numpy as np

from .weights import EncoderConfig, tensor_order


class UnsupportedModel(ValueError):
    """The directory describes a model the HIP encoder does not run (maps to MX_EUNSUPPORTED)."""


def _read_json(path: str) -> dict:
    with open(path, "r", encoding="utf-8") as f:
        return json.load(f)


def _load_state(path: str) -> Dict[str, np.ndarray]:
    st = os.path.join(path, "model.safetensors")
    if os.path.exists(st):
        from safetensors.numpy import load_file
        try:
            return dict(load_file(st))
        except Exception:  # bf16 tensors are not a numpy dtype: go through torch
            import torch  # noqa: F401
            from safetensors.torch import load_file as load_torch
            return {k: v.float().numpy() for k, v in load_torch(st).items()}
    pb = os.path.join(path, "pytorch_model.bin")
    if os.path.exists(pb):
        import torch
        sd = torch.load(pb, map_location="cpu", weights_only=True)
        return {k: v.float().numpy() for k, v in sd.items()}
    ot = os.path.join(path, "rust_model.ot")
    if os.path.exists(ot):
        # The file create_model() itself reads (embedding.rs:99-100): rust-bert's weights, written by tch's Tensor::save_multi =
        # libtorch's OutputArchive -- a TorchScript archive (zip: data.pkl + data/N) whose parameters / buffers carry the
        # checkpoint's tensor names with '.' spelled '|' (a TorchScript attribute name cannot hold a dot).  torch.jit.load reads
        # such an archive; VERIFIED HERE only against archives torch.jit.save wrote (tests/test_pretrained.py): no rust_model.ot can
        # be fetched offline.  Anything torch.jit.load refuses is refused with the file's name.
        try:
            import torch
            mod = torch.jit.load(ot, map_location="cpu")
            sd = {k.replace("|", "."): v.detach().float().numpy() for k, v in list(mod.named_parameters()) + list(mod.named_buffers())}
        except Exception as e:  # noqa: BLE001
            raise UnsupportedModel(f"{path}: rust_model.ot is not a TorchScript archive torch.jit.load reads ({type(e).__name__}); "
                                   "provide model.safetensors or pytorch_model.bin") from e
        if not sd:
            raise UnsupportedModel(f"{path}: rust_model.ot holds no tensors")
        return sd
    raise FileNotFoundError(f"{path}: no model.safetensors / pytorch_model.bin / rust_model.ot")


def default_precision(hidden: int, pooling: str) -> str:
    """The precision a loader picks when the caller names none: the bf16 ingest mode -- except for CLS-pooled hidden-768
    models (bge-base-en), whose scores BETWEEN embeddings move by 1e-2 ... 4e-2 on bf16 operands under checkpoint-like weights
    (one token, twelve layers; tests/test_encoder_gpu.py) against north_star's 1e-3 on scores: those get "bf16x3"
    (MX_PREC_BF16X3), the one mode that held the bar on every draw of such weights that was tried (<= 6e-4; the cheaper
    "mixed" reaches 1.1e-3 on two of eleven, profiles/r6_precision_modes_over_seeds.txt).  Mean pooling averages the rounding
    noise of hundreds of tokens (1.3e-3 at worst on the MiniLM shapes) and keeps bf16."""
    return "bf16x3" if pooling == "cls" and hidden == 768 else "bf16"


def load_pretrained_dir(path: str, precision: Optional[str] = None) -> Tuple[EncoderConfig, Dict[str, np.ndarray], Optional[str], dict]:
    """-> (EncoderConfig, tensors by HF name, path of vocab.txt or None, info).  ``precision``: "bf16" | "bf16x3" | "mixed" | "mixed1", or
    None = :func:`default_precision` of the model.

    ``info``: ``do_lower_case``, ``model_type``, ``modules`` (the pipeline's module types in order), ``bpe_files``
    (``(vocab.json, merges.txt)`` of a byte-level BPE tokenizer, or None), ``tokenizer_json`` (path of ``tokenizer.json``, or
    None: the fallback when neither ``vocab.txt`` nor the BPE pair is there).

    Raises :class:`UnsupportedModel`, ``OSError``, ``KeyError`` or ``ValueError`` -- what ``from_pretrained_dir`` turns into
    ``SetupError``; whatever a damaged file provokes underneath (a list where an object belongs, safetensors' own error type)
    is re-raised as ``ValueError``."""
    try:
        return _load_pretrained_dir(path, precision)
    except (UnsupportedModel, OSError, KeyError, ValueError):
        raise
    except Exception as e:  # noqa: BLE001
        raise ValueError(f"{path}: damaged model directory ({type(e).__name__}: {e})") from e


def _load_pretrained_dir(path: str, precision: Optional[str]):
    if not os.path.isdir(path):
        raise FileNotFoundError(path)
    modules_path = os.path.join(path, "modules.json")
    pooling_dir, normalize, transformer_dir, module_types = "1_Pooling", False, "", []
    if os.path.exists(modules_path):
        for m in sorted(_read_json(modules_path), key=lambda m: m.get("idx", 0)):
            ty = m.get("type", "").rsplit(".", 1)[-1]
            module_types.append(ty)
            if ty == "Transformer":
                transformer_dir = m.get("path", "")
            elif ty == "Pooling":
                pooling_dir = m.get("path", "1_Pooling")
            elif ty == "Normalize":
                normalize = True
            else:  # Dense, LayerNorm, WeightedLayerPooling, CNN, ... : not part of the HIP forward
                raise UnsupportedModel(f"{path}: module '{m.get('path', ty)}' ({m.get('type')}) is not supported")
    else:  # a bare transformers checkpoint: mean pooling + normalisation must then be asked for by the caller's config
        module_types = ["Transformer"]
    tdir = os.path.join(path, transformer_dir)
    hc = _read_json(os.path.join(tdir, "config.json"))
    mtype = hc.get("model_type", "bert")
    if mtype not in ("bert", "roberta", "xlm-roberta", "distilroberta"):
        raise UnsupportedModel(f"{path}: model_type '{mtype}' (BERT / RoBERTa encoder stacks only)")
    if hc.get("hidden_act", "gelu") != "gelu":
        raise UnsupportedModel(f"{path}: hidden_act '{hc.get('hidden_act')}' (erf GELU only)")
    if hc.get("position_embedding_type", "absolute") != "absolute":
        raise UnsupportedModel(f"{path}: position_embedding_type '{hc.get('position_embedding_type')}'")
    if hc.get("embedding_size", hc["hidden_size"]) != hc["hidden_size"]:
        raise UnsupportedModel(f"{path}: factorised embeddings (embedding_size != hidden_size)")
    roberta = mtype != "bert"
    pos_offset = (int(hc.get("pad_token_id", 1)) + 1) if roberta else 0

    sb_path = os.path.join(path, "sentence_bert_config.json")
    sb = _read_json(sb_path) if os.path.exists(sb_path) else {}
    max_seq = int(sb.get("max_seq_length") or min(512, hc["max_position_embeddings"] - pos_offset))

    pooling = "mean"
    pc_path = os.path.join(path, pooling_dir, "config.json")
    if os.path.exists(pc_path):
        pc = _read_json(pc_path)
        modes = [k for k in ("pooling_mode_cls_token", "pooling_mode_mean_tokens", "pooling_mode_max_tokens",
                             "pooling_mode_mean_sqrt_len_tokens", "pooling_mode_weightedmean_tokens", "pooling_mode_lasttoken")
                 if pc.get(k)]
        if modes == ["pooling_mode_cls_token"]:
            pooling = "cls"
        elif modes == ["pooling_mode_mean_tokens"]:
            pooling = "mean"
        else:
            raise UnsupportedModel(f"{path}: pooling modes {modes} (CLS or mean only)")
        if pc.get("word_embedding_dimension", hc["hidden_size"]) != hc["hidden_size"]:
            raise UnsupportedModel(f"{path}: pooling dimension {pc.get('word_embedding_dimension')} != hidden_size")

    cfg = EncoderConfig(layers=int(hc["num_hidden_layers"]), hidden=int(hc["hidden_size"]), heads=int(hc["num_attention_heads"]),
                        ffn=int(hc["intermediate_size"]), vocab=int(hc["vocab_size"]), max_pos=int(hc["max_position_embeddings"]),
                        type_vocab=int(hc.get("type_vocab_size", 2)), ln_eps=float(hc.get("layer_norm_eps", 1e-12)),
                        pooling=pooling, normalize=normalize, max_seq_length=max_seq, pos_offset=pos_offset,
                        precision=precision if precision is not None else default_precision(int(hc["hidden_size"]), pooling))

    state = _load_state(tdir)
    # keep what the encoder reads (drops pooler.*, position_ids, lm heads) and check the shapes now, not at upload time
    prefixes = ("", "bert.", "roberta.", "0.auto_model.", "model.", "auto_model.")
    tensors: Dict[str, np.ndarray] = {}
    for name, shape in tensor_order(cfg):
        t = next((state[p + name] for p in prefixes if p + name in state), None)
        if t is None:
            raise KeyError(f"{path}: weight '{name}' missing")
        a = np.asarray(t, dtype=np.float32)
        if tuple(a.shape) != shape:
            raise ValueError(f"{path}: {name} has shape {a.shape}, config.json implies {shape}")
        tensors[name] = a

    vocab = next((p for p in (os.path.join(path, "vocab.txt"), os.path.join(tdir, "vocab.txt")) if os.path.exists(p)), None)
    # RoBERTa-family checkpoints (all-distilroberta-v1) ship a byte-level BPE tokenizer instead: vocab.json + merges.txt
    bpe = next(((os.path.join(d_, "vocab.json"), os.path.join(d_, "merges.txt")) for d_ in (path, tdir)
                if os.path.exists(os.path.join(d_, "vocab.json")) and os.path.exists(os.path.join(d_, "merges.txt"))), None)
    lower = sb.get("do_lower_case")
    tc_path = os.path.join(path, "tokenizer_config.json")
    if os.path.exists(tc_path):
        lower = _read_json(tc_path).get("do_lower_case", lower)
    tj = next((p for p in (os.path.join(path, "tokenizer.json"), os.path.join(tdir, "tokenizer.json")) if os.path.exists(p)), None)
    info = {"do_lower_case": True if lower is None else bool(lower), "model_type": mtype, "modules": module_types,
            "bpe_files": bpe, "tokenizer_json": tj}
    return cfg, tensors, vocab, info
