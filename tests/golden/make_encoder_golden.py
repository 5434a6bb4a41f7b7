"""Cross-checks oracle/bert_oracle.py against transformers.BertModel (CPU, f32/f64) on seeded weights
and writes tests/golden/encoder_golden.npz (inputs as seeds, outputs as arrays).

Runs in the dev container only (transformers is an independent implementation of the same
published architecture, not the reference; the Rust reference cannot be built here).  Weights come
from memex_amd.weights.synthetic_weights(cfg, seed), regenerated identically on the GPU box.
    python tests/golden/make_encoder_golden.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from memex_amd.weights import EncoderConfig, synthetic_weights  # noqa: E402
from oracle import bert_oracle  # noqa: E402

CASES = {  # name: (cfg kwargs, B, S, seed)
    "l2_h384": (dict(layers=2, hidden=384, heads=12, ffn=1536, vocab=2000), 4, 32, 101),
    "l6_h384": (dict(layers=6, hidden=384, heads=12, ffn=1536, vocab=2000), 4, 128, 102),
    "l2_h768_cls": (dict(layers=2, hidden=768, heads=12, ffn=3072, vocab=2000, pooling="cls"), 4, 64, 103),
    # RoBERTa-style embeddings (all-distilroberta-v1, embedding.rs:29): checked against transformers.RobertaModel
    "l2_h768_roberta": (dict(layers=2, hidden=768, heads=12, ffn=3072, vocab=2000, max_pos=514, type_vocab=1,
                             ln_eps=1e-5, pos_offset=2), 4, 64, 104),
}


def inputs(cfg, B, S, seed):
    rng = np.random.default_rng(seed)
    ids = rng.integers(1000, cfg.vocab, size=(B, S)).astype(np.int32)
    lens = rng.integers(max(1, S // 4), S + 1, size=B).astype(np.int32)
    lens[0] = S
    lens[1] = 1
    return ids, lens


def hf_forward(cfg, w, ids, lens):
    from transformers import BertConfig, BertModel, RobertaConfig, RobertaModel
    roberta = cfg.pos_offset != 0
    Config, Model = (RobertaConfig, RobertaModel) if roberta else (BertConfig, BertModel)
    extra = dict(pad_token_id=cfg.pos_offset - 1) if roberta else {}   # position ids = padding_idx + 1 + t
    hc = Config(vocab_size=cfg.vocab, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers,
                    num_attention_heads=cfg.heads, intermediate_size=cfg.ffn, max_position_embeddings=cfg.max_pos,
                    type_vocab_size=cfg.type_vocab, layer_norm_eps=cfg.ln_eps, hidden_act="gelu",
                    hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **extra)
    m = Model(hc, add_pooling_layer=False).double().eval()
    sd = {k: torch.from_numpy(v).double() for k, v in w.items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not [k for k in missing if "position_ids" not in k], missing
    assert not unexpected, unexpected
    S = ids.shape[1]
    mask = (np.arange(S)[None, :] < lens[:, None]).astype(np.int64)
    with torch.no_grad():
        h = m(input_ids=torch.from_numpy(ids.astype(np.int64)), attention_mask=torch.from_numpy(mask)).last_hidden_state
    h = h.numpy()
    if cfg.pooling == "cls":
        pooled = h[:, 0]
    else:
        pooled = (h * mask[:, :, None]).sum(1) / np.maximum(mask.sum(1, keepdims=True), 1e-9)
    return pooled / np.maximum(np.linalg.norm(pooled, axis=1, keepdims=True), 1e-12)


if __name__ == "__main__":
    out = {}
    for name, (kw, B, S, seed) in CASES.items():
        cfg = EncoderConfig(**kw)
        w = synthetic_weights(cfg, seed)
        ids, lens = inputs(cfg, B, S, seed)
        ours = bert_oracle.encode(w, cfg.as_dict(), ids, lens)
        hf = hf_forward(cfg, w, ids, lens)
        d = np.abs(ours - hf).max()
        print(f"{name}: max |oracle - transformers| = {d:.3e}")
        assert d < 1e-9, "oracle restatement disagrees with transformers.BertModel"
        out[name + "_out"] = hf.astype(np.float64)
        out[name + "_meta"] = np.array([B, S, seed], dtype=np.int64)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "encoder_golden.npz"), **out)
    print("wrote", len(CASES), "cases")
