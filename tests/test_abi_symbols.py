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


def test_sharded_entries_without_gpu_fail_loudly(trag):
    """The sharded entry points resolve RCCL at run time and, like the store, have no CPU path."""
    import torch
    if torch.cuda.is_available():
        return
    from trueno_rag_amd import sharded
    assert sharded.rccl_version() >= 20000          # librccl was found and answers
    for make in (lambda: sharded.MultiGpuVectorStore(trag.VectorStoreConfig(dimension=4), [0], "copy"),
                 lambda: sharded.ShardGroup(0, 1, 0, 10)):
        try:
            make()
        except trag.TragError as e:
            assert e.status == 3 and "no HIP device" in str(e)
        else:
            raise AssertionError("a sharded object was created without a GPU")


def test_fuse_errors_carry_their_own_message(trag):
    """Every failing host entry sets the thread's last error (no stale message of an earlier call)."""
    import ctypes as C
    L = trag.lib()
    n = C.c_uint32(7)
    ids = (C.c_uint64 * 2)(1, 2)
    out_ids, out_sc = (C.c_uint64 * 4)(), (C.c_float * 4)()
    assert L.trag_fuse(9, 60.0, ids, None, 2, ids, None, 2, out_ids, out_sc, C.byref(n)) == 5
    assert b"unknown strategy" in L.trag_last_error() and n.value == 0
    assert L.trag_fuse(1, 0.5, ids, None, 2, ids, None, 2, out_ids, out_sc, C.byref(n)) == 5
    assert b"scores" in L.trag_last_error()
    assert L.trag_merge_topk(None, None, None, 2, 3, 3, None, None, C.byref(n)) == 5
    assert b"trag_merge_topk" in L.trag_last_error()
