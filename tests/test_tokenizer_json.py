"""tokenizer.json -> native tokenizer (mx_tokenizer_create_from_json, csrc/tokenizer.cpp): the one file
`Tokenizer::from_pretrained` reads in segment_text (reference lib/libmemex/src/llm/embedding.rs:163).  The files are
written here by the `tokenizers` package (the crate the reference links) from synthetic vocabularies -- the real ones are
not reachable offline -- and the native handle built from them must agree with the package AND with the native handle built
from the split files (vocab.txt / vocab.json + merges.txt).  No GPU needed."""
import json

import numpy as np
import pytest

from test_tokenizer import TEXTS as WP_TEXTS, make_vocab
from test_tokenizer_bpe import CORPUS, SPECIALS, TEXTS as BPE_TEXTS


@pytest.fixture(scope="module")
def wp(lib_built, tmp_path_factory):
    from tokenizers import BertWordPieceTokenizer
    from memex_amd.tokenizer import JsonTokenizer, WordPieceTokenizer
    d = tmp_path_factory.mktemp("wpjson")
    (d / "vocab.txt").write_text("\n".join(make_vocab()) + "\n", encoding="utf-8")
    hf = BertWordPieceTokenizer(str(d / "vocab.txt"), lowercase=True)
    # what the published all-MiniLM tokenizer.json carries besides the model: fixed padding + truncation to 128
    # (the reference's own test_tokenizer, embedding.rs:204-217, sees that padding: encoding.len() == 128)
    hf.enable_truncation(128)
    hf.enable_padding(length=128)
    hf.save(str(d / "tokenizer.json"))
    hf.no_truncation()
    hf.no_padding()
    return hf, JsonTokenizer(str(d / "tokenizer.json")), WordPieceTokenizer(str(d / "vocab.txt"), lowercase=True), d


def test_wordpiece_from_tokenizer_json(wp):
    hf, js, split, d = wp
    assert js.vocab == split.vocab == len(make_vocab())
    for t in WP_TEXTS:
        for special in (False, True):
            e = hf.encode(t, add_special_tokens=special)
            assert js.encode(t, special) == e.ids == split.encode(t, special), (t, special)
            assert js.decode(e.ids, True) == hf.decode(e.ids, skip_special_tokens=True)
    rng = np.random.default_rng(3)
    text = " ".join(rng.choice(["tax", "the", "Union", "don't", "unbelievable", "中文", "résumé", "zzzqqq", "."], size=900))
    assert js.windows(text, 256, 86) == split.windows(text, 256, 86)
    # from memory as well
    from memex_amd.tokenizer import JsonTokenizer
    mem = JsonTokenizer((d / "tokenizer.json").read_bytes())
    assert mem.encode(WP_TEXTS[1], True) == js.encode(WP_TEXTS[1], True)


def test_cased_tokenizer_json_keeps_case(lib_built, tmp_path):
    from tokenizers import BertWordPieceTokenizer
    from memex_amd.tokenizer import JsonTokenizer
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "The", "the", "Café", "cafe", "##s", "."]
    (tmp_path / "vocab.txt").write_text("\n".join(vocab) + "\n", encoding="utf-8")
    hf = BertWordPieceTokenizer(str(tmp_path / "vocab.txt"), lowercase=False)
    hf.save(str(tmp_path / "tokenizer.json"))
    js = JsonTokenizer(str(tmp_path / "tokenizer.json"))
    for t in ("The the Café cafe.", "THE Thes thes"):
        assert js.encode(t, True) == hf.encode(t).ids, t


@pytest.fixture(scope="module")
def bpe(lib_built, tmp_path_factory):
    from tokenizers import ByteLevelBPETokenizer
    from tokenizers.processors import RobertaProcessing
    from memex_amd.tokenizer import ByteLevelBpeTokenizer, JsonTokenizer
    d = tmp_path_factory.mktemp("bpejson")
    tr = ByteLevelBPETokenizer()
    tr.train_from_iterator(CORPUS, vocab_size=700, min_frequency=1, special_tokens=SPECIALS, show_progress=False)
    tr.save_model(str(d))
    hf = ByteLevelBPETokenizer(str(d / "vocab.json"), str(d / "merges.txt"))
    from tokenizers import AddedToken
    # (as the tokenizer.json of the RoBERTa family lists them: <mask> swallows the white space in front of it -- what the handle built
    # from vocab.json + merges.txt assumes, and what the JSON handle reads from `added_tokens`)
    hf.add_special_tokens(SPECIALS[:4] + [AddedToken("<mask>", lstrip=True, special=True)])
    hf._tokenizer.post_processor = RobertaProcessing(("</s>", hf.token_to_id("</s>")), ("<s>", hf.token_to_id("<s>")))
    hf.save(str(d / "tokenizer.json"))
    return hf, JsonTokenizer(str(d / "tokenizer.json")), ByteLevelBpeTokenizer(str(d / "vocab.json"), str(d / "merges.txt")), d


def test_byte_level_bpe_from_tokenizer_json(bpe):
    hf, js, split, d = bpe
    assert js.vocab == split.vocab
    for t in BPE_TEXTS:
        for special in (False, True):
            e = hf.encode(t, add_special_tokens=special)
            assert js.encode(t, special) == e.ids == split.encode(t, special), (t, special)
            assert js.decode(e.ids, True) == hf.decode(e.ids, skip_special_tokens=True)
    text = " ".join(np.random.default_rng(5).choice(["tax", "Biden", "don't", "résumé", "привет", "中文", "🙂", "2023"], size=600))
    assert js.windows(text, 256, 86) == split.windows(text, 256, 86)


def test_merges_as_strings_and_as_pairs(bpe):
    """tokenizers < 0.20 wrote merges as "a b" strings, later versions as ["a", "b"] pairs: both forms load."""
    from memex_amd.tokenizer import JsonTokenizer
    hf, js, split, d = bpe
    doc = json.loads((d / "tokenizer.json").read_text(encoding="utf-8"))
    m = doc["model"]["merges"]
    other = [" ".join(x) for x in m] if isinstance(m[0], list) else [x.split(" ") for x in m]
    doc["model"]["merges"] = other
    alt = JsonTokenizer(json.dumps(doc).encode("utf-8"))
    alt2 = JsonTokenizer(json.dumps(doc, ensure_ascii=True, indent=1).encode("utf-8"))
    for t in BPE_TEXTS:
        assert alt.encode(t, True) == js.encode(t, True) == alt2.encode(t, True), t


def test_unsupported_and_damaged_files_are_refused(wp, bpe, tmp_path):
    from memex_amd._lib import MX_EINVAL, MX_EIO, MX_EUNSUPPORTED, MemexHipError
    from memex_amd.tokenizer import JsonTokenizer
    base = json.loads((wp[3] / "tokenizer.json").read_text(encoding="utf-8"))

    def code_of(doc):
        with pytest.raises(MemexHipError) as ei:
            JsonTokenizer(doc if isinstance(doc, bytes) else json.dumps(doc).encode("utf-8"))
        return ei.value.code

    for edit in (lambda d: d["normalizer"].update(type="NFKC"), lambda d: d.update(normalizer=None),
                 lambda d: d["pre_tokenizer"].update(type="Whitespace"), lambda d: d["model"].update(type="Unigram"),
                 lambda d: d["model"].update(continuing_subword_prefix="@@"), lambda d: d["normalizer"].update(strip_accents=False),
                 lambda d: d["normalizer"].update(handle_chinese_chars=False), lambda d: d["model"].update(max_input_chars_per_word=20)):
        doc = json.loads(json.dumps(base))
        edit(doc)
        assert code_of(doc) == MX_EUNSUPPORTED
    b = json.loads((bpe[3] / "tokenizer.json").read_text(encoding="utf-8"))
    for edit in (lambda d: d["pre_tokenizer"].update(add_prefix_space=True), lambda d: d["model"].update(dropout=0.1),
                 lambda d: d["model"].update(byte_fallback=True), lambda d: d.update(normalizer={"type": "Lowercase"})):
        doc = json.loads(json.dumps(b))
        edit(doc)
        assert code_of(doc) == MX_EUNSUPPORTED
    raw = (wp[3] / "tokenizer.json").read_bytes()
    for bad in (b"", b"[1, 2]", b"{}", b'{"model": {"type": "WordPiece"}}', raw[: len(raw) // 2], raw.replace(b'"[UNK]": 1', b'"[UNK]": -1'),
                b'{"model": ' + b"[" * 200 + b"]" * 200 + b"}"):
        assert code_of(bad) == MX_EINVAL, bad[:40]
    with pytest.raises(MemexHipError) as ei:
        JsonTokenizer(str(tmp_path / "missing.json"))
    assert ei.value.code == MX_EIO and "Unable to load model" in ei.value.msg
    rng = np.random.default_rng(0)
    for _ in range(200):  # random damage: an error code or a working handle, never a crash
        a = bytearray(raw)
        for _ in range(int(rng.integers(1, 6))):
            i = int(rng.integers(0, len(a)))
            a[i:i + int(rng.integers(0, 40))] = bytes(rng.integers(0, 256, size=int(rng.integers(0, 8)), dtype=np.uint8))
        try:
            JsonTokenizer(bytes(a)).encode("the tax", True)
        except MemexHipError:
            pass


def test_pretrained_dir_with_only_tokenizer_json(wp, tmp_path):
    """A model directory that ships tokenizer.json alone (no vocab.txt): the loaders fall back to it."""
    from memex_amd import embedding as E
    tok = E._as_tokenizer(str(wp[3] / "tokenizer.json"), 0)
    assert type(tok).__name__ == "JsonTokenizer" and tok.encode("the tax", True) == wp[0].encode("the tax").ids
