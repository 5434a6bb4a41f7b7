"""Native WordPiece tokenizer / segmenter vs the `tokenizers` Python package (same Rust core the
reference links, lib/libmemex/Cargo.toml:31) on a synthetic vocabulary -- no GPU needed.

The reference's own tokenizer test (embedding.rs:204-217) needs the HF hub; what it pins -- that a
short string encodes without error under truncation 256 / stride 128 -- is restated at the end."""
import numpy as np
import pytest

SPECIALS = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
STEMS = ["the", "of", "and", "tax", "taxes", "bid", "en", "biden", "say", "s", "about", "what", "do", "does", "not",
         "work", "ing", "ed", "er", "est", "un", "believ", "able", "cafe", "resume", "naive", "zurich", "state",
         "union", "2023", "20", "23", "a", "b", "c", "d", "e", "i", "o", "u", "x", "y", "z", "n", "t", "m", "re", "ve",
         "ll", "hello", "world", "token", "ize", "long", "word", "embed", "vector", "search", "gpu", "über", "uber"]
PUNCT = list("!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~")


def make_vocab():
    toks = list(SPECIALS)
    for s in STEMS:
        toks.append(s)
    for s in STEMS:
        toks.append("##" + s)
    toks += PUNCT
    toks += ["中", "文", "##s"]
    seen, out = set(), []
    for t in toks:
        if t not in seen:
            seen.add(t)
            out.append(t)
    return out


@pytest.fixture(scope="module")
def toks(lib_built, tmp_path_factory):
    from tokenizers import BertWordPieceTokenizer
    from memex_amd.tokenizer import WordPieceTokenizer
    vocab = make_vocab()
    p = tmp_path_factory.mktemp("vocab") / "vocab.txt"
    p.write_text("\n".join(vocab) + "\n", encoding="utf-8")
    hf = BertWordPieceTokenizer(str(p), lowercase=True)
    return hf, WordPieceTokenizer(str(p), lowercase=True), vocab


TEXTS = [
    "What does Biden say about taxes?",
    "The STATE of the Union 2023 -- unbelievable, isn't it?!",
    "Café résumé naïve Zürich ÜBER",
    "tokenizing   long\twords\nand don't re-embed; they've said: \"we'll do it\".",
    "unknownword zzzqqq the",
    "a" * 120 + " the",
    "中文 and the",
    "",
    "   ",
    "it 's the tax . do not say ' no ' !",
    # added tokens are cut out of the RAW text first, case-sensitively, whatever the normaliser does later
    "hello [SEP] world", "[CLS]x[MASK] [mask] [PAD][PAD]", "a [UNK] b [SE P] [SEP", "[SEP][SEP]the[CLS]",
]


def test_ids_match_hf_tokenizers(toks):
    hf, mine, _ = toks
    for t in TEXTS:
        for special in (False, True):
            assert mine.encode(t, special) == hf.encode(t, add_special_tokens=special).ids, (t, special)


def test_random_text_ids_and_decode(toks):
    hf, mine, _ = toks
    rng = np.random.default_rng(0)
    pieces = STEMS + PUNCT + ["Biden's", "WORKING", "unbelievable", "worked,", "x-y", "Ünion"]
    for _ in range(200):
        n = int(rng.integers(1, 40))
        t = " ".join(rng.choice(pieces, size=n))
        e = hf.encode(t, add_special_tokens=False)
        ids = mine.encode(t)
        assert ids == e.ids, t
        assert mine.decode(ids, True) == hf.decode(e.ids, skip_special_tokens=True), t


def test_segment_text_windows_match_reference_calls(toks):
    """The exact call sequence of embedding.rs:173-195 on the HF side vs mx_tokenizer_segment."""
    hf, mine, _ = toks
    rng = np.random.default_rng(1)
    text = " ".join(rng.choice(STEMS + ["don't", "it's", "' quoted '", "end."], size=900))
    for max_length, stride in ((256, 86), (256, 128), (64, 10), (1000, 86)):
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
    assert ids.shape == (3, 128) and lens.tolist() == [len(hf.encode(texts[0]).ids), 3, 128]
    cls, sep, pad = vocab.index("[CLS]"), vocab.index("[SEP]"), vocab.index("[PAD]")
    for b in range(3):
        assert ids[b, 0] == cls and ids[b, lens[b] - 1] == sep and (ids[b, lens[b]:] == pad).all()
    assert ids[0, : lens[0]].tolist() == hf.encode(texts[0]).ids
    short, slens = mine.encode_batch(texts[:2], 128)
    assert short.shape[1] == slens.max()                      # padded to the batch maximum only


def test_encode_batch_threads_match_single_texts(toks):
    """Large batches are split over host threads: every row must equal the one-text encoding."""
    import time
    hf, mine, _ = toks
    rng = np.random.default_rng(5)
    pieces = STEMS + PUNCT + ["Biden's", "WORKING", "unbelievable", "Ünion", "中文"]
    texts = [" ".join(pieces[int(i)] for i in rng.integers(0, len(pieces), int(rng.integers(1, 400)))) for _ in range(700)]
    t0 = time.perf_counter()
    ids, lens = mine.encode_batch(texts, 256)
    dt = time.perf_counter() - t0
    assert ids.shape[0] == 700 and lens.max() <= 256
    for b in range(0, 700, 7):
        want = mine.encode(texts[b], False)[:254]
        assert ids[b, 1: lens[b] - 1].tolist() == want
    assert int(lens.sum()) / dt > 1e5          # sanity only (tokens/s); a hang or serial fallback shows up in CI time


def test_reference_tokenizer_test_shape(toks):
    """embedding.rs:204-217: a 5-word string encodes under truncation(256, stride 128) (the reference
    then sees 128 ids only because the downloaded tokenizer.json pads to a fixed 128)."""
    hf, mine, _ = toks
    ids = mine.encode("this is a test string")
    assert 0 < len(ids) <= 256 and mine.windows("this is a test string", 256, 128) == [mine.decode(ids, True)]


def test_errors(lib_built):
    from memex_amd import _lib
    from memex_amd.tokenizer import WordPieceTokenizer
    with pytest.raises(_lib.MemexHipError) as ei:
        WordPieceTokenizer("/nonexistent/vocab.txt")
    assert ei.value.code == _lib.MX_EIO
    with pytest.raises(_lib.MemexHipError):
        WordPieceTokenizer(["a", "b"])                       # no special tokens
    t = WordPieceTokenizer(make_vocab())
    with pytest.raises(_lib.MemexHipError):
        t.windows("x", 10, 10)                               # stride must be < max_length


# ---- full-Unicode normaliser / pre-tokenizer (generated tables, scripts/gen_unicode_tables.py) ----------
UNI_STEMS = ["привет", "мир", "при", "##вет", "ελληνικα", "ελλη", "##νικα", "γεια", "σου", "ς", "σ", "istanbul", "i",
             "strasse", "ß", "ss", "ᄒ", "ᅡ", "ᆫ", "한", "##ᅡ", "##ᆫ", "नमसत", "न", "##म", "##स", "##त", "عربي", "ع", "##ر",
             "##ب", "##ي", "𝓐", "ａ", "ｂ", "fi", "ﬁ", "dz", "ǆ", "ǳ", "あ", "か", "##か", "カ", "ガ", "é", "e", "œ", "ø", "đ",
             "ł", "ı", "ŉ", "ʼ", "n", "##n", "x", "##x", "🙂", "🫠", "ก", "ิ", "##ิ"]
UNI_PUNCT = ["¡", "¿", "«", "»", "–", "—", "‘", "’", "“", "”", "…", "、", "。", "「", "」", "！", "？", "·", "§", "¶", "‰",
             "₀", "+", "÷", "×", "©", "™", "°", "^", "~", "|", "¦", "¬", "$", "€", "£"]


@pytest.fixture(scope="module")
def uni_toks(lib_built, tmp_path_factory):
    from tokenizers import BertWordPieceTokenizer
    from memex_amd.tokenizer import WordPieceTokenizer
    seen, vocab = set(), []
    for t in SPECIALS + STEMS + ["##" + s for s in STEMS] + PUNCT + UNI_STEMS + UNI_PUNCT:
        if t not in seen:
            seen.add(t)
            vocab.append(t)
    p = tmp_path_factory.mktemp("vocab_uni") / "vocab.txt"
    p.write_text("\n".join(vocab) + "\n", encoding="utf-8")
    out = {}
    for lower in (True, False):
        out[lower] = (BertWordPieceTokenizer(str(p), lowercase=lower, strip_accents=None),
                      WordPieceTokenizer(str(p), lowercase=lower))
    return out


UNI_TEXTS = [
    "ПРИВЕТ, Мир!  Привет—мир…", "ΕΛΛΗΝΙΚΑ Γειά σου ΣΟΥ Σ σ ς", "İstanbul I ı İ i̇", "Straße STRASSE ẞ ß",
    "한국어 한 한", "नमस्ते नमस्ते।", "عَرَبِيّ ﻋﺮﺑﻲ", "ｆｕｌｌ ｗｉｄｔｈ ａｂ Ａ", "ﬁ ﬂ ǅ ǲ Ǆ", "がか ガカ が",
    "é é É É ñ ñ å å ç ç", "Œuvre Øre Đà Łódź ŉ ʼn", "x­y x​y x‍y x﻿y x y x y x　y",
    "🙂 🫠 x🙂y", "กิ กิ", "à̖b à̖b", "„Zitat“ «cita» ‹x› 「引用」 a·b a‧b 5‰ 1÷2×3", "tab\there\r\nnew\x0bline\x0cfeed\x1funit\x7fdel",
    "한".encode("utf-8", "ignore").decode(), "ǅemal Ǉ ǈ ǉ ǋ", "ΐ ΰ ẖ ǰ ﬃ ﬆ ք ﬓ", "K Å Ω", "ſ ẛ ς",
]


def test_unicode_texts_match_hf(uni_toks):
    for lower, (hf, mine) in uni_toks.items():
        for t in UNI_TEXTS:
            assert mine.encode(t, False) == hf.encode(t, add_special_tokens=False).ids, (lower, t)


def test_unicode_fuzz_matches_hf(uni_toks):
    """Random strings drawn from many scripts / categories plus the vocabulary's own characters: the
    native normaliser + pre-tokenizer + WordPiece must agree with `tokenizers` id for id."""
    rng = np.random.default_rng(1234)
    blocks = [(0x20, 0x7f), (0xa0, 0x17f), (0x180, 0x24f), (0x250, 0x2ff), (0x300, 0x36f), (0x370, 0x3ff), (0x400, 0x52f),
              (0x530, 0x58f), (0x590, 0x6ff), (0x900, 0x97f), (0xe00, 0xe7f), (0x10a0, 0x10ff), (0x1100, 0x11ff),
              (0x1e00, 0x1fff), (0x2000, 0x206f), (0x2070, 0x20cf), (0x2100, 0x218f), (0x2190, 0x22ff), (0x2460, 0x24ff),
              (0x2c00, 0x2dff), (0x3000, 0x30ff), (0x3130, 0x318f), (0x4e00, 0x4e80), (0xa640, 0xa69f), (0xac00, 0xad00),
              (0xd7a0, 0xd7ff), (0xfb00, 0xfb4f), (0xfe00, 0xfe6f), (0xff00, 0xffef), (0x10400, 0x1044f), (0x1d400, 0x1d4ff),
              (0x1e900, 0x1e95f), (0x1f300, 0x1f64f), (0x1fa70, 0x1faff), (0xe0000, 0xe007f), (0x0, 0x1f), (0x7f, 0x9f)]
    own = sorted({ch for t in UNI_STEMS + UNI_PUNCT + STEMS for ch in t.replace("##", "")})
    for lower, (hf, mine) in uni_toks.items():
        for it in range(1500):
            n = int(rng.integers(1, 24))
            chars = []
            for _ in range(n):
                r = rng.random()
                if r < 0.35:
                    chars.append(own[int(rng.integers(0, len(own)))])
                elif r < 0.5:
                    chars.append(" ")
                else:
                    a, b = blocks[int(rng.integers(0, len(blocks)))]
                    c = int(rng.integers(a, b + 1))
                    if 0xd800 <= c <= 0xdfff or c == 0:
                        c = 0x41
                    chars.append(chr(c))
            t = "".join(chars)
            want = hf.encode(t, add_special_tokens=False).ids
            got = mine.encode(t, False)
            assert got == want, (lower, [hex(ord(c)) for c in t])


# ---- the one-pass encoder (hash-table WordPiece, prepared decoder strings) and the document batch ------------------------
def _big_vocab(rng):
    """A BERT-sized vocabulary of word-like stems and ## continuations over a small alphabet (so that random words
    split into several pieces and some fail to split at all)."""
    letters = list("etaoinshrdlcumwfgypbvkjxqz")
    word = lambda n: "".join(rng.choice(letters, size=n))
    stems, conts = set(), set()
    while len(stems) < 12000:
        stems.add(word(int(rng.integers(1, 8))))
    while len(conts) < 6000:
        conts.add("##" + word(int(rng.integers(1, 5))))
    toks = SPECIALS + letters[:20] + ["##" + c for c in letters[:23]] + sorted(stems) + sorted(conts) + PUNCT + \
        [str(i) for i in range(10)] + ["é", "über", "##ß", "中"]
    seen, out = set(), []
    for t in toks:
        if t not in seen:
            seen.add(t)
            out.append(t)
    return out, sorted(stems), letters


@pytest.fixture(scope="module")
def big_toks(lib_built, tmp_path_factory):
    from tokenizers import BertWordPieceTokenizer
    from memex_amd.tokenizer import WordPieceTokenizer
    rng = np.random.default_rng(77)
    vocab, stems, letters = _big_vocab(rng)
    p = tmp_path_factory.mktemp("vocab_big") / "vocab.txt"
    p.write_text("\n".join(vocab) + "\n", encoding="utf-8")
    return BertWordPieceTokenizer(str(p), lowercase=True), WordPieceTokenizer(str(p), lowercase=True), stems, letters


def _doc(rng, stems, letters, chars):
    ws, n = [], 0
    while n < chars:
        r = rng.random()
        if r < 0.6:
            w = stems[int(rng.zipf(1.3)) % len(stems)]
        elif r < 0.9:
            w = "".join(rng.choice(letters, size=int(rng.integers(3, 14))))     # splits into pieces, or [UNK]
        elif r < 0.93:
            w = "".join(rng.choice(letters, size=int(rng.integers(95, 108))))   # around max_input_chars_per_word = 100
        else:
            w = str(rng.choice([",", ".", "'s", "--", "(x)", "2023", "Über", "café", "don't", "中文", "a­b", "\t", "\n\n"]))
        if rng.random() < 0.1:
            w = w.capitalize()
        ws.append(w)
        n += len(w) + 1
    return " ".join(ws)


def test_one_pass_encoder_matches_staged_and_hf_on_a_bert_sized_vocabulary(big_toks):
    hf, mine, stems, letters = big_toks
    rng = np.random.default_rng(3)
    for _ in range(30):
        t = _doc(rng, stems, letters, int(rng.integers(10, 4000)))
        ids = mine.encode(t)
        assert ids == mine.encode_staged(t)
        assert ids == hf.encode(t, add_special_tokens=False).ids
        assert mine.decode(ids, True) == hf.decode(ids, skip_special_tokens=True)
    # the 100-code-point limit: 100 letters are still split, 101 are [UNK] (also when the characters are multi-byte)
    for ch in ("e", "é"):
        for n in (99, 100, 101):
            w = ch * n
            assert mine.encode(w) == mine.encode_staged(w) == hf.encode(w, add_special_tokens=False).ids, (ch, n)


def test_one_pass_encoder_matches_staged_on_unicode_fuzz(uni_toks):
    rng = np.random.default_rng(99)
    for lower, (_, mine) in uni_toks.items():
        for _ in range(3000):
            n = int(rng.integers(1, 40))
            cps = [int(rng.integers(0x20, 0x3000)) if rng.random() < 0.7 else int(rng.integers(0x3000, 0x1fb00)) for _ in range(n)]
            t = "".join(chr(c) for c in cps if not 0xd800 <= c <= 0xdfff)
            assert mine.encode(t) == mine.encode_staged(t), (lower, [hex(c) for c in cps])
        for raw in (b"ab\xffcd \xe2\x82 x", b"\xc3", b"x\xf0\x9f\x99y", b"\x80\x80 a", b"a\xed\xa0\x80b"):   # malformed UTF-8: both forms drop the same bytes
            import ctypes
            from memex_amd._lib import check, lib
            outs = []
            for fn, extra in ((lib().mx_tokenizer_encode, (0,)), (lib().mx_tokenizer_encode_staged, ())):
                ids = (ctypes.c_int32 * 64)()
                k = ctypes.c_int(0)
                check(fn(mine._h, raw, *extra, ids, 64, ctypes.byref(k)))
                outs.append(list(ids[: k.value]))
            assert outs[0] == outs[1], raw


def test_windows_batch_equals_one_document_at_a_time(big_toks):
    """mx_tokenizer_segment_batch deals documents to host threads: every document's windows must be the reference call
    sequence's (embedding.rs:173-195), whatever its neighbours."""
    hf, mine, stems, letters = big_toks
    rng = np.random.default_rng(11)
    docs = [_doc(rng, stems, letters, int(rng.integers(0, 30000))) for _ in range(40)] + ["", "   ", "x", "' a ' b"]
    got = mine.windows_batch(docs, 256, 86)
    assert len(got) == len(docs)
    hf.enable_truncation(max_length=256, stride=86)
    for d, w in zip(docs, got):
        enc = hf.encode(d, add_special_tokens=False)
        want = [hf.decode(enc.ids, skip_special_tokens=True).replace(" ' ", "'")]
        want += [hf.decode(o.ids, skip_special_tokens=True) for o in enc.overflowing]
        assert w == want
    hf.no_truncation()
    assert mine.windows_batch([], 256, 86) == []
    # a text whose windows outgrow the first buffer estimate (every character becomes "c " + overlap): the call is repeated
    dense = "." * 5000
    assert mine.windows_batch([dense], 8, 7)[0] == mine.windows(dense, 8, 7) and len(mine.windows(dense, 8, 7)) == 4993
