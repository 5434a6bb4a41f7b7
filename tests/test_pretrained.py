"""Loading what the reference loads (embedding.rs:99-100): a sentence-transformers directory -> EncoderConfig + tensors +
vocab.  The directory is written here by `transformers` (``BertModel(config).save_pretrained``, random init -- pretrained
checkpoints are not reachable offline) plus the sentence-transformers side files, then read back by
``memex_amd.pretrained.load_pretrained_dir``; on the GPU box the embedder built from it is compared with the f64 oracle fed
the same tensors and the same token ids."""
import json
import os

import numpy as np
import pytest

from test_tokenizer import make_vocab


def make_st_dir(root, *, hidden=384, layers=2, heads=12, ffn=1536, pooling="mean", normalize=True, max_seq_length=128,
                weights="safetensors", dense=False, hidden_act="gelu", pooling_modes=None, seed=0):
    """A sentence-transformers model directory with a random-init BERT of the given shape and a synthetic vocabulary."""
    import torch
    from transformers import BertConfig, BertModel
    vocab = make_vocab()
    torch.manual_seed(seed)
    hc = BertConfig(vocab_size=len(vocab), hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads,
                    intermediate_size=ffn, max_position_embeddings=512, hidden_act=hidden_act)
    model = BertModel(hc, add_pooling_layer=False).eval()
    with torch.no_grad():  # trained-looking LayerNorms / biases instead of the ones / zeros of a fresh init
        for n, p in model.named_parameters():
            if n.endswith("LayerNorm.weight"):
                p.copy_(1.0 + 0.1 * torch.randn_like(p))
            elif n.endswith(".bias"):
                p.copy_(0.05 * torch.randn_like(p))
            elif "embeddings" in n:
                p.copy_(0.05 * torch.randn_like(p))
            else:
                p.copy_(torch.randn_like(p) / p.shape[1] ** 0.5)
    os.makedirs(root, exist_ok=True)
    model.save_pretrained(root, safe_serialization=(weights == "safetensors"))
    if weights == "bin" and not os.path.exists(os.path.join(root, "pytorch_model.bin")):
        torch.save(model.state_dict(), os.path.join(root, "pytorch_model.bin"))
        for f in ("model.safetensors",):
            if os.path.exists(os.path.join(root, f)):
                os.remove(os.path.join(root, f))
    with open(os.path.join(root, "vocab.txt"), "w", encoding="utf-8") as f:
        f.write("\n".join(vocab) + "\n")
    mods = [{"idx": 0, "name": "0", "path": "", "type": "sentence_transformers.models.Transformer"},
            {"idx": 1, "name": "1", "path": "1_Pooling", "type": "sentence_transformers.models.Pooling"}]
    if dense:
        mods.append({"idx": 2, "name": "2", "path": "2_Dense", "type": "sentence_transformers.models.Dense"})
    if normalize:
        mods.append({"idx": len(mods), "name": str(len(mods)), "path": f"{len(mods)}_Normalize", "type": "sentence_transformers.models.Normalize"})
    json.dump(mods, open(os.path.join(root, "modules.json"), "w"))
    json.dump({"max_seq_length": max_seq_length, "do_lower_case": False}, open(os.path.join(root, "sentence_bert_config.json"), "w"))
    json.dump({"do_lower_case": True, "tokenizer_class": "BertTokenizer"}, open(os.path.join(root, "tokenizer_config.json"), "w"))
    os.makedirs(os.path.join(root, "1_Pooling"), exist_ok=True)
    pm = pooling_modes or {"pooling_mode_cls_token": pooling == "cls", "pooling_mode_mean_tokens": pooling == "mean",
                           "pooling_mode_max_tokens": False, "pooling_mode_mean_sqrt_len_tokens": False}
    json.dump({"word_embedding_dimension": hidden, **pm}, open(os.path.join(root, "1_Pooling", "config.json"), "w"))
    return model, vocab


@pytest.mark.parametrize("weights", ["safetensors", "bin"])
def test_directory_round_trip(tmp_path, weights):
    from memex_amd.pretrained import load_pretrained_dir
    from memex_amd.weights import pack_weights, tensor_order
    d = str(tmp_path / "st")
    model, vocab = make_st_dir(d, hidden=384, layers=2, pooling="cls", max_seq_length=200, weights=weights)
    cfg, tensors, vpath, info = load_pretrained_dir(d)
    assert (cfg.layers, cfg.hidden, cfg.heads, cfg.ffn, cfg.vocab, cfg.max_pos, cfg.type_vocab) == (2, 384, 12, 1536, len(vocab), 512, 2)
    assert cfg.pooling == "cls" and cfg.normalize is True and cfg.max_seq_length == 200 and cfg.pos_offset == 0
    assert abs(cfg.ln_eps - 1e-12) < 1e-18 and cfg.precision == "bf16"
    assert vpath == os.path.join(d, "vocab.txt") and info["do_lower_case"] is True and info["modules"] == ["Transformer", "Pooling", "Normalize"]
    sd = {k: v.detach().numpy() for k, v in model.state_dict().items()}
    for name, shape in tensor_order(cfg):
        np.testing.assert_array_equal(tensors[name], sd[name])
    blob = pack_weights(tensors, cfg)
    assert blob.dtype == np.float32 and blob.size == sum(int(np.prod(s)) for _, s in tensor_order(cfg))
    assert load_pretrained_dir(d, precision="bf16x3")[0].precision == "bf16x3"


def test_rust_model_ot_is_read_when_it_is_a_torchscript_archive(tmp_path):
    """The weight file create_model() itself reads (embedding.rs:99-100) is rust_model.ot: tch's save_multi = libtorch's
    OutputArchive, a TorchScript archive whose tensors carry the checkpoint's names with '.' spelled '|'.  No such file can be
    fetched offline, so the archive is written HERE (torch.jit.save of a module holding the same tensors under those names): the
    loader reads it through torch.jit.load and hands out the tensors of model.safetensors bit for bit."""
    import torch
    from memex_amd.pretrained import load_pretrained_dir
    d = str(tmp_path / "ot")
    model, _ = make_st_dir(d, hidden=384, layers=2, weights="safetensors")
    want = load_pretrained_dir(d, precision="bf16")[1]

    class Holder(torch.nn.Module):
        def forward(self):
            return 0

    holder = Holder()
    for name, tensor in model.state_dict().items():
        holder.register_buffer(("bert." + name).replace(".", "|"), tensor.clone())   # rust-bert's names carry the model prefix
    torch.jit.save(torch.jit.script(holder), os.path.join(d, "rust_model.ot"))
    os.remove(os.path.join(d, "model.safetensors"))
    cfg, got, _, _ = load_pretrained_dir(d, precision="bf16")
    from memex_amd.weights import pack_weights
    np.testing.assert_array_equal(pack_weights(got, cfg), pack_weights(want, cfg))


def test_loaders_choose_the_bar_meeting_precision_for_cls_pooled_hidden_768(tmp_path):
    """a CLS-pooled hidden-768 model (bge-base-en) moves its scores by up to 1e-2 on bf16 operands (north_star: 1e-3);
    a loader that is not told otherwise picks MX_PREC_BF16X3 for it (<= 6e-4 on every weight seed tried: tests/test_encoder_gpu.py), bf16 for everything else."""
    from memex_amd.pretrained import default_precision, load_pretrained_dir
    assert default_precision(768, "cls") == "bf16x3" and default_precision(768, "mean") == "bf16" and default_precision(384, "cls") == "bf16"
    d = str(tmp_path / "bge")
    make_st_dir(d, hidden=768, layers=1, pooling="cls", weights="safetensors")
    assert load_pretrained_dir(d)[0].precision == "bf16x3"
    assert load_pretrained_dir(d, precision="bf16")[0].precision == "bf16"
    d2 = str(tmp_path / "mean768")
    make_st_dir(d2, hidden=768, layers=1, pooling="mean", weights="safetensors")
    assert load_pretrained_dir(d2)[0].precision == "bf16"


def test_unsupported_models_are_refused(tmp_path):
    """What the HIP encoder does not implement must fail loudly (MX_EUNSUPPORTED's Python face), not be approximated."""
    from memex_amd.pretrained import UnsupportedModel, load_pretrained_dir
    make_st_dir(str(tmp_path / "dense"), layers=1, dense=True)
    with pytest.raises(UnsupportedModel, match="2_Dense"):
        load_pretrained_dir(str(tmp_path / "dense"))
    make_st_dir(str(tmp_path / "maxpool"), layers=1, pooling_modes={"pooling_mode_cls_token": False, "pooling_mode_mean_tokens": False,
                                                                     "pooling_mode_max_tokens": True})
    with pytest.raises(UnsupportedModel, match="pooling"):
        load_pretrained_dir(str(tmp_path / "maxpool"))
    make_st_dir(str(tmp_path / "relu"), layers=1, hidden_act="relu")
    with pytest.raises(UnsupportedModel, match="hidden_act"):
        load_pretrained_dir(str(tmp_path / "relu"))
    make_st_dir(str(tmp_path / "ot"), layers=1)
    os.remove(str(tmp_path / "ot" / "model.safetensors"))
    open(str(tmp_path / "ot" / "rust_model.ot"), "wb").write(b"PK")
    with pytest.raises(UnsupportedModel, match="rust_model.ot"):
        load_pretrained_dir(str(tmp_path / "ot"))
    with pytest.raises(FileNotFoundError):
        load_pretrained_dir(str(tmp_path / "nowhere"))
    # a tensor whose shape contradicts config.json
    make_st_dir(str(tmp_path / "bad"), layers=1)
    hc = json.load(open(str(tmp_path / "bad" / "config.json")))
    hc["intermediate_size"] = 768
    json.dump(hc, open(str(tmp_path / "bad" / "config.json"), "w"))
    with pytest.raises(ValueError, match="shape"):
        load_pretrained_dir(str(tmp_path / "bad"))


def test_cpp_loader_agrees_with_the_python_one(tmp_path, lib_built):
    """include/memex_pretrained.hpp (the C++ host mirror's loader): same configuration, same weight blob byte for byte,
    same refusals as memex_amd.pretrained."""
    import subprocess
    from memex_amd.pretrained import load_pretrained_dir
    from memex_amd.weights import pack_weights
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "test_pretrained")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "test_pretrained.cpp"), "-o", exe, "-L", os.path.join(root, "memex_amd"),
                           "-lmemex_hip", "-lpthread", "-Wl,-rpath," + os.path.join(root, "memex_amd")])
    d = str(tmp_path / "st")
    make_st_dir(d, hidden=384, layers=2, pooling="cls", max_seq_length=200)
    r = subprocess.run([exe, d, "1"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("OK "), r.stdout + r.stderr
    f = r.stdout.split()
    cfg, tensors, vpath, info = load_pretrained_dir(d, precision="bf16x3")
    blob = pack_weights(tensors, cfg)
    with np.errstate(over="ignore"):  # position-weighted sum of the 32-bit words, mod 2^64 (as the C++ side computes it)
        h = int((blob.view(np.uint32).astype(np.uint64) * np.arange(1, blob.size + 1, dtype=np.uint64)).sum(dtype=np.uint64))
    assert [int(v) for v in f[1:8]] == [cfg.layers, cfg.hidden, cfg.heads, cfg.ffn, cfg.vocab, cfg.max_pos, cfg.type_vocab]
    assert abs(float(f[8]) - cfg.ln_eps) < 1e-15 and [int(v) for v in f[9:13]] == [1, 1, 0, 1]      # cls, normalize, pos_offset, bf16x3
    assert int(f[13]) == cfg.max_seq_length == 200 and int(f[14]) == 1 and int(f[15]) == blob.nbytes
    assert f[16] == f"{h:016x}" and f[17] == vpath
    for name, kw in (("dense", dict(dense=True)), ("relu", dict(hidden_act="relu")),
                     ("maxpool", dict(pooling_modes={"pooling_mode_cls_token": False, "pooling_mode_mean_tokens": False, "pooling_mode_max_tokens": True}))):
        make_st_dir(str(tmp_path / name), layers=1, **kw)
        r = subprocess.run([exe, str(tmp_path / name)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 3 and r.stdout.startswith("REFUSED"), (name, r.stdout + r.stderr)
    make_st_dir(str(tmp_path / "bin"), layers=1, weights="bin")
    r = subprocess.run([exe, str(tmp_path / "bin")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "safetensors" in r.stdout


def test_loaders_survive_damaged_files(tmp_path, lib_built):
    """Bit flips, truncations and insertions in config.json / modules.json / 1_Pooling/config.json and in the safetensors header:
    both loaders either load the directory or refuse it with their own error -- never a crash (C++: exit code 0 or 3, no
    signal, no foreign exception) and never an exception type the callers do not catch (Python)."""
    import shutil
    import struct
    import subprocess
    from memex_amd.pretrained import UnsupportedModel, load_pretrained_dir
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "test_pretrained")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "test_pretrained.cpp"), "-o", exe, "-L", os.path.join(root, "memex_amd"),
                           "-lmemex_hip", "-lpthread", "-Wl,-rpath," + os.path.join(root, "memex_amd")])
    good = str(tmp_path / "good")
    make_st_dir(good, hidden=32, heads=2, ffn=64, layers=1)
    rng = np.random.default_rng(5)
    outcomes = {"cpp": {0: 0, 3: 0}, "py": {"ok": 0, "refused": 0}}
    for it in range(120):
        d = str(tmp_path / f"m{it}")
        shutil.copytree(good, d)
        target = ["config.json", "modules.json", os.path.join("1_Pooling", "config.json"), "model.safetensors",
                  "sentence_bert_config.json"][it % 5]
        raw = bytearray(open(os.path.join(d, target), "rb").read())
        if target == "model.safetensors":
            hl = struct.unpack("<Q", raw[:8])[0]
            lo, hi = (0, 8 + hl) if it % 10 != 3 else (0, 8)              # the header (sometimes its length word only)
        else:
            lo, hi = 0, len(raw)
        kind = int(rng.integers(0, 4))
        if kind == 0:                                                      # flip a few bytes
            for _ in range(int(rng.integers(1, 4))):
                raw[int(rng.integers(lo, hi))] = int(rng.integers(0, 256))
        elif kind == 1:                                                    # truncate
            del raw[int(rng.integers(lo, hi)):]
        elif kind == 2:                                                    # insert structural noise
            at = int(rng.integers(lo, hi))
            raw[at:at] = rng.choice([b"[[[[", b'"\\u12', b"{", b"}", b"-", b"1e999", b'"\\', b"[" * 300, b"null"])
        else:                                                              # swap a digit (shapes, offsets, sizes)
            digits = [i for i in range(lo, hi) if 48 <= raw[i] <= 57]
            if digits:
                raw[digits[int(rng.integers(0, len(digits)))]] = 48 + int(rng.integers(0, 10))
        open(os.path.join(d, target), "wb").write(bytes(raw))
        r = subprocess.run([exe, d], capture_output=True, text=True, timeout=60)
        assert r.returncode in (0, 3), (it, target, kind, r.returncode, r.stdout[-300:], r.stderr[-300:])
        outcomes["cpp"][r.returncode] += 1
        try:
            load_pretrained_dir(d)
            outcomes["py"]["ok"] += 1
        except (UnsupportedModel, OSError, KeyError, ValueError):          # what SentenceEmbedder.from_pretrained_dir turns into SetupError
            outcomes["py"]["refused"] += 1
        shutil.rmtree(d)
    assert outcomes["cpp"][3] >= 20 and outcomes["cpp"][0] >= 5, outcomes   # the mutations bite, and harmless ones still load


@pytest.mark.gpu
@pytest.mark.parametrize("hidden,heads,ffn,pooling", [(384, 12, 1536, "mean"), (768, 12, 3072, "cls")])
def test_embedder_from_pretrained_dir_matches_the_oracle(tmp_path, lib_built, hidden, heads, ffn, pooling):
    """SentenceEmbedder.from_pretrained_dir: config + weights + native tokenizer from the directory; embeddings within 1e-3
    of the f64 oracle fed the same tensors and the ids the HF tokenizer produces for the same vocabulary."""
    from tokenizers import BertWordPieceTokenizer
    from memex_amd import embedding as E
    from memex_amd.pretrained import load_pretrained_dir
    from oracle import bert_oracle
    d = str(tmp_path / "st")
    make_st_dir(d, hidden=hidden, layers=3, heads=heads, ffn=ffn, pooling=pooling, max_seq_length=64, seed=5)
    texts = ["What does Biden say about taxes?", "The STATE of the Union 2023 -- unbelievable, isn't it?!",
             "tokenizing long words and don't re-embed; they've said: \"we'll do it\".", "hello world " * 60]
    th, emb = E.SentenceEmbedder.from_pretrained_dir(d)
    got = np.asarray([emb.encode_single(t).vector for t in texts], dtype=np.float64)
    segs = emb.encode("the tax state union " * 200)                 # segment_text through the directory's vocabulary
    emb.shutdown()
    cfg, tensors, vpath, _ = load_pretrained_dir(d)
    hf = BertWordPieceTokenizer(vpath, lowercase=True)
    hf.enable_truncation(max_length=cfg.max_seq_length)
    encs = [hf.encode(t) for t in texts]
    S = max(len(e.ids) for e in encs)
    ids = np.zeros((len(texts), S), dtype=np.int32)
    lens = np.asarray([len(e.ids) for e in encs], dtype=np.int32)
    for i, e in enumerate(encs):
        ids[i, :len(e.ids)] = e.ids
    assert lens.max() == 64                                           # the long text hit max_seq_length
    ref = bert_oracle.encode(tensors, cfg.as_dict(), ids, lens)
    cos = (got * ref).sum(1) / np.linalg.norm(got, axis=1) / np.linalg.norm(ref, axis=1)
    assert (1.0 - cos).max() <= 1e-3, cos
    assert len(segs) == 1 + int(np.ceil((800 - 256) / 170)) and all(len(s.vector) == hidden for s in segs)
    with pytest.raises(E.SetupError):
        make_st_dir(str(tmp_path / "dense"), layers=1, dense=True)
        E.SentenceEmbedder.from_pretrained_dir(str(tmp_path / "dense"))
    # the same directory in the split-operand mode: f32-grade agreement with the oracle
    th, emb = E.SentenceEmbedder.from_pretrained_dir(d, precision="bf16x3")
    got3 = np.asarray([emb.encode_single(t).vector for t in texts], dtype=np.float64)
    emb.shutdown()
    cos3 = (got3 * ref).sum(1) / np.linalg.norm(got3, axis=1) / np.linalg.norm(ref, axis=1)
    assert (1.0 - cos3).max() <= 1e-7, cos3


def make_roberta_dir(root, *, hidden=768, layers=2, heads=12, ffn=3072, max_seq_length=64, seed=0):
    """A sentence-transformers directory of the RoBERTa family (all-distilroberta-v1's shape): RobertaModel weights,
    a byte-level BPE tokenizer trained here (vocab.json + merges.txt), mean pooling + normalise."""
    import torch
    from tokenizers import ByteLevelBPETokenizer
    from transformers import RobertaConfig, RobertaModel
    from test_tokenizer_bpe import CORPUS, SPECIALS
    os.makedirs(root, exist_ok=True)
    tr = ByteLevelBPETokenizer()
    tr.train_from_iterator(CORPUS, vocab_size=700, min_frequency=1, special_tokens=SPECIALS, show_progress=False)
    tr.save_model(root)
    vocab = json.load(open(os.path.join(root, "vocab.json"), encoding="utf-8"))
    torch.manual_seed(seed)
    hc = RobertaConfig(vocab_size=len(vocab), hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads,
                       intermediate_size=ffn, max_position_embeddings=514, type_vocab_size=1, layer_norm_eps=1e-5,
                       pad_token_id=vocab["<pad>"], bos_token_id=vocab["<s>"], eos_token_id=vocab["</s>"])
    model = RobertaModel(hc, add_pooling_layer=False).eval()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("LayerNorm.weight"):
                p.copy_(1.0 + 0.1 * torch.randn_like(p))
            elif n.endswith(".bias"):
                p.copy_(0.05 * torch.randn_like(p))
            elif "embeddings" in n:
                p.copy_(0.05 * torch.randn_like(p))
            else:
                p.copy_(torch.randn_like(p) / p.shape[1] ** 0.5)
    model.save_pretrained(root, safe_serialization=True)
    json.dump([{"idx": 0, "name": "0", "path": "", "type": "sentence_transformers.models.Transformer"},
               {"idx": 1, "name": "1", "path": "1_Pooling", "type": "sentence_transformers.models.Pooling"},
               {"idx": 2, "name": "2", "path": "2_Normalize", "type": "sentence_transformers.models.Normalize"}],
              open(os.path.join(root, "modules.json"), "w"))
    json.dump({"max_seq_length": max_seq_length, "do_lower_case": False}, open(os.path.join(root, "sentence_bert_config.json"), "w"))
    os.makedirs(os.path.join(root, "1_Pooling"), exist_ok=True)
    json.dump({"word_embedding_dimension": hidden, "pooling_mode_cls_token": False, "pooling_mode_mean_tokens": True,
               "pooling_mode_max_tokens": False, "pooling_mode_mean_sqrt_len_tokens": False},
              open(os.path.join(root, "1_Pooling", "config.json"), "w"))
    return model, vocab


def test_roberta_directory_is_recognised(tmp_path):
    from memex_amd.pretrained import load_pretrained_dir
    d = str(tmp_path / "rb")
    model, vocab = make_roberta_dir(d, layers=1)
    cfg, tensors, vpath, info = load_pretrained_dir(d)
    assert (cfg.hidden, cfg.max_pos, cfg.type_vocab, cfg.pos_offset, cfg.pooling, cfg.normalize) == (768, 514, 1, 2, "mean", True)
    assert abs(cfg.ln_eps - 1e-5) < 1e-12 and vpath is None and info["model_type"] == "roberta"
    assert info["bpe_files"] == (os.path.join(d, "vocab.json"), os.path.join(d, "merges.txt"))
    np.testing.assert_array_equal(tensors["embeddings.position_embeddings.weight"], model.state_dict()["embeddings.position_embeddings.weight"].numpy())


@pytest.mark.gpu
def test_roberta_embedder_from_pretrained_dir_matches_the_oracle(tmp_path, lib_built):
    """The third model segment_text accepts (all-distilroberta-v1's shape): RoBERTa-style embeddings (pos_offset 2) + the native
    byte-level BPE tokenizer, end to end from the directory; the oracle gets the same tensors and the ids `tokenizers` produces."""
    from tokenizers import ByteLevelBPETokenizer
    from tokenizers.processors import RobertaProcessing
    from memex_amd import embedding as E
    from memex_amd.pretrained import load_pretrained_dir
    from oracle import bert_oracle
    d = str(tmp_path / "rb")
    make_roberta_dir(d, layers=3, max_seq_length=48, seed=6)
    texts = ["What does Biden say about taxes?", "Café résumé naïve Zürich ÜBER -- 🙂!", "don't re-embed; they've said: \"we'll do it\".",
             "the tax " * 80]
    th, emb = E.SentenceEmbedder.from_pretrained_dir(d, E.ModelConfig(model=E.EmbeddingsModelType.AllDistilrobertaV1))
    got = np.asarray([emb.encode_single(t).vector for t in texts], dtype=np.float64)
    segs = emb.encode("the tax state union " * 150)
    emb.shutdown()
    cfg, tensors, _, info = load_pretrained_dir(d)
    hf = ByteLevelBPETokenizer(*info["bpe_files"])
    hf._tokenizer.post_processor = RobertaProcessing(("</s>", hf.token_to_id("</s>")), ("<s>", hf.token_to_id("<s>")))
    hf.enable_truncation(max_length=cfg.max_seq_length)
    encs = [hf.encode(t) for t in texts]
    S = max(len(e.ids) for e in encs)
    ids = np.full((len(texts), S), hf.token_to_id("<pad>"), dtype=np.int32)
    lens = np.asarray([len(e.ids) for e in encs], dtype=np.int32)
    for i, e in enumerate(encs):
        ids[i, :len(e.ids)] = e.ids
    assert lens.max() == 48
    ref = bert_oracle.encode(tensors, cfg.as_dict(), ids, lens)
    cos = (got * ref).sum(1) / np.linalg.norm(got, axis=1) / np.linalg.norm(ref, axis=1)
    assert (1.0 - cos).max() <= 1e-3, cos
    assert len(segs) >= 2 and all(len(s.vector) == 768 for s in segs)
