"""Native byte-level BPE tokenizer / segmenter (mx_tokenizer_create_bpe, csrc/tokenizer.cpp) vs the `tokenizers` Python
package (the Rust crate the reference links, lib/libmemex/Cargo.toml:31) -- the tokenizer of all-distilroberta-v1, the
third model `segment_text` accepts (embedding.rs:159).  No GPU needed.  The real vocab.json / merges.txt are not reachable
offline, so a byte-level BPE vocabulary is TRAINED here by the `tokenizers` package on synthetic text (a few hundred merges
over Latin, accented, Cyrillic, CJK and emoji material) and both implementations load the files it saves."""
import json
import os

import numpy as np
import pytest

CORPUS = [
    "What does Biden say about taxes? The State of the Union 2023 -- unbelievable, isn't it?!",
    "tokenizing long words and don't re-embed; they've said: \"we'll do it\". I'm sure he'd agree, you're right.",
    "Café résumé naïve Zürich ÜBER straße Œuvre Łódź ñandú", "Привет мир, привет! как дела 123 456.78",
    "中文 分词 测试 日本語 テスト 한국어", "emoji 🙂 🫠 🎉 mixed x🙂y", "numbers 1 22 333 4444 3.14159 1,000,000 2023-09-29",
    "tabs\tand\nnewlines\r\n  and   runs    of spaces ", "the the the of of and and tax taxes taxing taxed",
    "embedding vector search gpu kernel matrix core bandwidth roofline", "a b c d e f g h i j k l m n o p q r s t u v w x y z",
    "'s 't 're 've 'm 'll 'd 'S 'T it's IT'S rock'n'roll o'clock", "under_score snake_case camelCase kebab-case path/to/file.txt",
] * 4
SPECIALS = ["<s>", "<pad>", "</s>", "<unk>", "<mask>"]

TEXTS = [
    "What does Biden say about taxes?", "The STATE of the Union 2023 -- unbelievable, isn't it?!", "Café résumé naïve Zürich ÜBER",
    "tokenizing   long\twords\nand don't re-embed; they've said: \"we'll do it\".", "unknownword zzzqqq the", "a" * 120 + " the",
    "中文 and the", "", "   ", " leading and trailing  ", "\n\n\nnew\n\nlines\n", "it 's the tax . do not say ' no ' !",
    "x y z　w", "I'M SHOUTING, AREN'T I? we'd've", "🙂🙂 🫠", "1e10 3.5% $100 #tag @user", "tab\t\tend\t",
    # added tokens are cut out of the RAW text first: "<s>" is an HTML tag, and a document may hold any of these
    "a<s>b", "x <mask> y", "x  \t<mask>y <mask>", "</s></s>", "<pad> <unk>x", "a <mas k> < mask> <MASK> <s", "<s><s>the</s>",
]


@pytest.fixture(scope="module")
def toks(lib_built, tmp_path_factory):
    from tokenizers import ByteLevelBPETokenizer
    from tokenizers.processors import RobertaProcessing
    from memex_amd.tokenizer import ByteLevelBpeTokenizer
    d = tmp_path_factory.mktemp("bpe")
    trainer = ByteLevelBPETokenizer()
    trainer.train_from_iterator(CORPUS, vocab_size=700, min_frequency=1, special_tokens=SPECIALS, show_progress=False)
    trainer.save_model(str(d))
    vj, mg = str(d / "vocab.json"), str(d / "merges.txt")
    hf = ByteLevelBPETokenizer(vj, mg)
    from tokenizers import AddedToken
    # (as the tokenizer.json of the RoBERTa family lists them: <mask> swallows the white space in front of it)
    hf.add_special_tokens(SPECIALS[:4] + [AddedToken("<mask>", lstrip=True, special=True)])
    hf._tokenizer.post_processor = RobertaProcessing(("</s>", hf.token_to_id("</s>")), ("<s>", hf.token_to_id("<s>")))
    return hf, ByteLevelBpeTokenizer(vj, mg), json.load(open(vj, encoding="utf-8"))


def test_ids_and_decode_match_hf(toks):
    hf, mine, vocab = toks
    assert mine.vocab == len(vocab)
    for t in TEXTS:
        for special in (False, True):
            e = hf.encode(t, add_special_tokens=special)
            ids = mine.encode(t, special)
            assert ids == e.ids, (t, special)
            for skip in (True, False):
                assert mine.decode(ids, skip) == hf.decode(e.ids, skip_special_tokens=skip), (t, special, skip)


def test_segment_text_windows_match_reference_calls(toks):
    """embedding.rs:173-195 on the HF side vs mx_tokenizer_segment: windows that cut multi-byte characters decode through
    the same lossy UTF-8."""
    hf, mine, _ = toks
    rng = np.random.default_rng(1)
    words = ["tax", "Biden", "don't", "it's", "' quoted '", "end.", "résumé", "привет", "中文", "🙂", "2023", "   ", "\n"]
    text = " ".join(rng.choice(words, size=700))
    for max_length, stride in ((256, 86), (64, 10), (7, 3), (3000, 86)):
        hf.enable_truncation(max_length=max_length, stride=stride)
        enc = hf.encode(text, add_special_tokens=False)
        want = [hf.decode(enc.ids, skip_special_tokens=True).replace(" ' ", "'")]
        want += [hf.decode(o.ids, skip_special_tokens=True) for o in enc.overflowing]
        hf.no_truncation()
        got = mine.windows(text, max_length, stride)
        assert got == want, (max_length, stride, len(got), len(want))
    assert mine.windows("", 256, 86) == [""]


def test_encode_batch_for_the_encoder(toks):
    hf, mine, vocab = toks
    texts = ["what does biden say about taxes?", "the", " ".join(["tax"] * 300)]
    ids, lens = mine.encode_batch(texts, 128)
    assert ids.shape == (3, 128) and lens.tolist() == [len(hf.encode(texts[0]).ids), len(hf.encode("the").ids), 128]
    for b in range(3):
        assert ids[b, 0] == vocab["<s>"] and ids[b, lens[b] - 1] == vocab["</s>"] and (ids[b, lens[b]:] == vocab["<pad>"]).all()
    assert ids[0, : lens[0]].tolist() == hf.encode(texts[0]).ids


def test_unicode_fuzz_matches_hf(toks):
    """Random strings over many scripts, the contraction patterns and whitespace runs: the hand-written matcher of the GPT-2
    regex + BPE must agree with `tokenizers` id for id, and the decoders text for text (skip_special on and off)."""
    hf, mine, _ = toks
    rng = np.random.default_rng(4321)
    blocks = [(0x20, 0x7f), (0xa0, 0x17f), (0x250, 0x36f), (0x370, 0x52f), (0x590, 0x6ff), (0x900, 0x97f), (0xe00, 0xe7f),
              (0x1100, 0x11ff), (0x1e00, 0x1fff), (0x2000, 0x206f), (0x2070, 0x218f), (0x2190, 0x22ff), (0x2460, 0x24ff),
              (0x3000, 0x30ff), (0x4e00, 0x4e80), (0xac00, 0xad00), (0xfb00, 0xfb4f), (0xff00, 0xffef), (0x10400, 0x1044f),
              (0x1d400, 0x1d7ff), (0x1f300, 0x1f64f), (0xe0000, 0xe007f), (0x0, 0x1f), (0x7f, 0x9f), (0x16a0, 0x16ff),
              (0x2150, 0x218f), (0x3040, 0x309f), (0xa8e0, 0xa8ff)]
    bits = ["'s", "'t", "'re", "'ve", "'m", "'ll", "'d", "'", " ", "  ", "\n", "\t", " \n ", "a", "B", "9", "42", ".", "!?", "é", "the"]
    for it in range(3000):
        n = int(rng.integers(1, 20))
        parts = []
        for _ in range(n):
            r = rng.random()
            if r < 0.45:
                parts.append(bits[int(rng.integers(0, len(bits)))])
            else:
                a, b = blocks[int(rng.integers(0, len(blocks)))]
                c = int(rng.integers(a, b + 1))
                if 0xd800 <= c <= 0xdfff or c == 0:
                    c = 0x41
                parts.append(chr(c))
        t = "".join(parts)
        e = hf.encode(t, add_special_tokens=False)
        got = mine.encode(t, False)
        assert got == e.ids, [hex(ord(c)) for c in t]
        assert mine.decode(got, True) == hf.decode(e.ids, skip_special_tokens=True), [hex(ord(c)) for c in t]
        # a prefix of the ids may end inside a multi-byte character: lossy decoding must agree too
        k = int(rng.integers(0, len(got) + 1))
        assert mine.decode(got[:k], True) == hf.decode(e.ids[:k], skip_special_tokens=True), ([hex(ord(c)) for c in t], k)


def test_errors(lib_built, tmp_path):
    from memex_amd import _lib
    from memex_amd.tokenizer import ByteLevelBpeTokenizer
    with pytest.raises(_lib.MemexHipError) as ei:
        ByteLevelBpeTokenizer("/nonexistent/vocab.json", "/nonexistent/merges.txt")
    assert ei.value.code == _lib.MX_EIO
    (tmp_path / "vocab.json").write_text('{"a": 0, "b": 1}')
    (tmp_path / "merges.txt").write_text("#version: 0.2\na b\n")
    with pytest.raises(_lib.MemexHipError):
        ByteLevelBpeTokenizer(str(tmp_path / "vocab.json"), str(tmp_path / "merges.txt"))       # no <s> </s> <pad>
    (tmp_path / "bad.json").write_text('["not", "an", "object"]')
    with pytest.raises(_lib.MemexHipError):
        ByteLevelBpeTokenizer(str(tmp_path / "bad.json"), str(tmp_path / "merges.txt"))
