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


@pytest.mark.gpu
def test_cpp_reference_tests(tmp_path, lib_built):
    exe = _build(tmp_path, lib_built)
    r = subprocess.run([exe, str(tmp_path / "work")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK 6 tests" in r.stdout
