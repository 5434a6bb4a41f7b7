"""memex_amd -- MI355X (gfx950) embedding + vector-search path for memex.

Everything here sits on ``libmemex_hip.so`` (hand-written HIP behind the C ABI of
``include/memex_hip.h``).  There is no CPU fallback: without the built library or a GPU the
calls raise.  Layout:

* ``_lib``      ctypes binding + in-tree build of the shared library
* ``index``     ``FlatIndex``: object wrapper over ``mx_index_*``
* ``storage``   mirror of the reference's ``VectorStore`` / ``HnswStore`` / ``get_vector_storage``
* ``weights``   encoder configs, HF-name weight packing, seeded synthetic weights
* ``pretrained`` a local sentence-transformers directory -> config + tensors + vocab (what rust-bert downloads)
* ``embedding`` mirror of the reference's ``SentenceEmbedder`` actor over ``mx_encoder_*``
* ``tasks``     the two callers of the path (worker ingest, API search) with the reference's segment ids
* ``sharded``   row-sharded multi-GPU index (one process per GPU, RCCL all-gather merge)
"""
from ._lib import MemexHipError, build, device_count, lib  # noqa: F401

__all__ = ["MemexHipError", "build", "device_count", "lib"]
