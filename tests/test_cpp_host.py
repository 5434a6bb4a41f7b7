"""The C++ host mirror (include/memex_hip.hpp) -- the reference is compiled code, so the host side
above the C ABI exists in C++ too.  CPU: it compiles against the header and links the library.
GPU: tests/cpp/test_store.cpp restates the reference's own tests (storage/local.rs:168-243)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path, lib_built):
    exe = str(tmp_path / "test_store")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_store.cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "memex_amd"), "-lmemex_hip", "-lpthread",
                           "-Wl,-rpath," + os.path.join(ROOT, "memex_amd")])
    return exe


def test_cpp_mirror_compiles_and_links(tmp_path, lib_built):
    assert os.path.exists(_build(tmp_path, lib_built))


def test_tail_weight_stream_layout(tmp_path, lib_built):
    """encoder_tail.hip's per-wave weight streams: a permutation of Wo, W1, W2 with every fragment where the
    kernel's addressing expects it (host code inside the library: runs without a GPU)."""
    exe = str(tmp_path / "test_tail_stream")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", os.path.join(ROOT, "tests", "cpp", "test_tail_stream.cpp"),
                           "-o", exe, "-L", os.path.join(ROOT, "memex_amd"), "-lmemex_hip", "-lpthread",
                           "-Wl,-rpath," + os.path.join(ROOT, "memex_amd")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "OK tail stream layout" in r.stdout, r.stdout + r.stderr


def test_shard_pool_hand_off(tmp_path):
    """The persistent helper threads of the sharded index (memex_amd/csrc/shard_pool.h): plain C++, no GPU."""
    exe = str(tmp_path / "test_shard_pool")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-pthread", os.path.join(ROOT, "tests", "cpp", "test_shard_pool.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "OK shard pool" in r.stdout, r.stdout + r.stderr


def test_debug_flag_parser(tmp_path):
    """MEMEX_HIP_DEBUG (memex_amd/csrc/mx_debug.h), the library's one kernel-variant switch: re-read when the string
    changes, defaults for absent keys, malformed items ignored, safe from several threads."""
    exe = str(tmp_path / "test_debug_flag")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-pthread", os.path.join(ROOT, "tests", "cpp", "test_debug_flag.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "OK debug flag" in r.stdout, r.stdout + r.stderr


def test_cpp_tokenizer_class_matches_the_tokenizers_package(tmp_path, lib_built):
    """memex::Tokenizer (the C++ host's face of mx_tokenizer_*): ids, decoded text and segment_text windows of a few
    documents equal the `tokenizers` package's (the crate the reference calls, embedding.rs:163-195)."""
    import numpy as np
    from tokenizers import BertWordPieceTokenizer
    from test_tokenizer import STEMS, make_vocab
    exe = str(tmp_path / "test_tokenizer_host")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_tokenizer_host.cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "memex_amd"), "-lmemex_hip", "-lpthread",
                           "-Wl,-rpath," + os.path.join(ROOT, "memex_amd")])
    vocab = tmp_path / "vocab.txt"
    vocab.write_text("\n".join(make_vocab()) + "\n", encoding="utf-8")
    rng = np.random.default_rng(8)
    pieces = STEMS + ["don't", "it's", "' quoted '", "end.", "Biden's", "Ünion", "中文", "x-y"]
    docs = [" ".join(rng.choice(pieces, size=int(n))) for n in (3, 40, 300, 900, 1500)] + ["", "the"]
    (tmp_path / "docs.txt").write_text("\n".join(docs) + "\n", encoding="utf-8")
    r = subprocess.run([exe, str(vocab), str(tmp_path / "docs.txt")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "OK tokenizer host" in r.stdout, r.stdout + r.stderr
    # the same tokenizer from the one-file form, tokenizer.json (Tokenizer::from_file): the same output, line for line
    BertWordPieceTokenizer(str(vocab), lowercase=True).save(str(tmp_path / "tokenizer.json"))
    rj = subprocess.run([exe, str(tmp_path / "tokenizer.json"), str(tmp_path / "docs.txt")], capture_output=True, text=True, timeout=120)
    assert rj.returncode == 0 and rj.stdout == r.stdout, rj.stdout + rj.stderr

    def fnv(b: bytes) -> int:
        h = 1469598103934665603
        for c in b:
            h = ((h ^ c) * 1099511628211) & (2 ** 64 - 1)
        return h
    hf = BertWordPieceTokenizer(str(vocab), lowercase=True)
    lines = [ln.split() for ln in r.stdout.splitlines() if ln.startswith("DOC")]
    assert len(lines) == len(docs)
    total = 0
    for d, ln in zip(docs, lines):
        hf.no_truncation()
        ids = hf.encode(d, add_special_tokens=False).ids
        hi = 1469598103934665603
        for i in ids:
            hi = ((hi ^ i) * 1099511628211) & (2 ** 64 - 1)
        hf.enable_truncation(max_length=256, stride=86)
        enc = hf.encode(d, add_special_tokens=False)
        wins = [hf.decode(enc.ids, skip_special_tokens=True).replace(" ' ", "'")] + \
            [hf.decode(o.ids, skip_special_tokens=True) for o in enc.overflowing]
        hw = 0
        for w in wins:
            hw = (hw * 31 + fnv(w.encode("utf-8"))) & (2 ** 64 - 1)
        assert int(ln[3]) == len(ids) and int(ln[4], 16) == hi, d[:40]
        assert int(ln[6], 16) == fnv(hf.decode(ids, skip_special_tokens=True).encode("utf-8"))
        assert int(ln[8]) == len(wins) and int(ln[9], 16) == hw
        total += min(len(ids), 126) + 2
    import uuid
    from memex_amd import tasks
    for ln in (l_.split() for l_ in r.stdout.splitlines() if l_.startswith("UUID ")):    # tasks.rs:36-40, db/document.rs:74
        doc = tasks.document_uuid(int(ln[1]))
        assert ln[2:] == [doc, tasks.segment_uuid(doc, 0), tasks.segment_uuid(doc, 71)]
    known = [l_.split()[1] for l_ in r.stdout.splitlines() if l_.startswith("UUID5 ")]
    assert known == [str(uuid.uuid5(uuid.NAMESPACE_DNS, "www.example.org"))]
    batch = [ln.split() for ln in r.stdout.splitlines() if ln.startswith("BATCH")][0]
    assert int(batch[2]) == 128 and int(batch[4]) == len(docs) and int(batch[6]) == total


@pytest.mark.gpu
def test_cpp_reference_tests(tmp_path, lib_built):
    exe = _build(tmp_path, lib_built)
    r = subprocess.run([exe, str(tmp_path / "work")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK 6 tests" in r.stdout
