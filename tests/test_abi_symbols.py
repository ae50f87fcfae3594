"""CPU: libtrag.so loads and exports every symbol include/trag.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "trag.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(trag_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_surface():
    syms = declared_symbols()
    for must in ("trag_store_create", "trag_store_insert", "trag_store_search", "trag_store_search_batch",
                 "trag_store_remove", "trag_store_len", "trag_fuse_rrf", "trag_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol(trag):
    lib = ctypes.CDLL(trag.lib_path())
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"declared in trag.h but not exported: {missing}"


def test_python_binding_covers_header(trag):
    from trueno_rag_amd import _abi
    assert sorted(_abi.SIGNATURES) == declared_symbols()


def test_no_gpu_is_reported_not_faked(trag):
    """Without a GPU, creating a store must fail loudly (no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        return
    try:
        trag.VectorStore.with_dimension(3)
    except trag.TragError as e:
        assert e.status == 3 and "no HIP device" in str(e)
    else:
        raise AssertionError("store creation succeeded without a GPU")
