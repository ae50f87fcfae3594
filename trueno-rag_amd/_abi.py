"""ctypes declarations for every symbol of include/trag.h."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# TRAG_LIB names another build of the same library (kernel experiments); never a fallback
_LIB = os.environ.get("TRAG_LIB") or os.path.join(_HERE, "lib", "libtrag.so")

OK, DIM_MISMATCH, INVALID_CONFIG, DEVICE_ERROR, OOM, INVALID_ARG, UNSUPPORTED = range(7)


class TragError(RuntimeError):
    """Error::VectorStore(String) and everything without a dedicated variant."""

    def __init__(self, status, message):
        super().__init__(f"trag status {status}: {message}")
        self.status = status


class DimensionMismatch(TragError):
    """Error::DimensionMismatch { expected, actual } (src/error.rs of the reference)."""

    def __init__(self, expected, actual, message=""):
        super().__init__(DIM_MISMATCH, message or f"dimension mismatch: expected {expected}, got {actual}")
        self.expected = expected
        self.actual = actual


class InvalidConfig(TragError):
    """Error::InvalidConfig(String)."""

    def __init__(self, message):
        super().__init__(INVALID_CONFIG, message)


class Config(C.Structure):
    _fields_ = [("dim", C.c_uint32), ("metric", C.c_uint32), ("device", C.c_int32),
                ("scan_mode", C.c_uint32), ("reserve_rows", C.c_uint64)]


class BM25Config(C.Structure):
    _fields_ = [("k1", C.c_float), ("b", C.c_float), ("device", C.c_int32)]


class ShardConfig(C.Structure):
    _fields_ = [("rank", C.c_uint32), ("world", C.c_uint32), ("device", C.c_int32), ("k_max", C.c_uint32),
                ("max_queries", C.c_uint32), ("overlap", C.c_uint32)]


SHARD_ID_BYTES = 128


class SearchStats(C.Structure):
    _fields_ = [("mode_used", C.c_uint32), ("candidates", C.c_uint32),
                ("refine_levels", C.c_uint32), ("fallbacks", C.c_uint32), ("keys_kept_from", C.c_uint32)]


# name -> (restype, argtypes); the single source the symbol-export test iterates over.
_fp, _u32p, _u64p = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
_H = C.c_void_p
SIGNATURES = {
    "trag_store_create": (C.c_int, [C.POINTER(Config), C.POINTER(_H)]),
    "trag_store_destroy": (None, [_H]),
    "trag_store_clone": (C.c_int, [_H, C.POINTER(_H)]),
    "trag_store_config": (C.c_int, [_H, C.POINTER(Config)]),
    "trag_store_insert": (C.c_int, [_H, _fp, C.c_uint64, C.c_uint32, _u32p]),
    "trag_store_overwrite": (C.c_int, [_H, C.c_uint32, _fp, C.c_uint32]),
    "trag_store_remove": (C.c_int, [_H, C.c_uint32]),
    "trag_store_len": (C.c_uint64, [_H]),
    "trag_store_rows": (C.c_uint64, [_H]),
    "trag_store_reserve": (C.c_int, [_H, C.c_uint64]),
    "trag_store_search": (C.c_int, [_H, _fp, C.c_uint32, C.c_uint32, _u32p, _fp, _u32p]),
    "trag_store_search_batch": (C.c_int, [_H, _fp, C.c_uint32, C.c_uint32, C.c_uint32, _u32p, _fp, _u32p]),
    "trag_store_search_device": (C.c_int, [_H, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "trag_store_set_pipelined": (C.c_int, [_H, C.c_int]),
    "trag_store_fence": (C.c_int, [_H, C.c_void_p]),
    "trag_store_search_device_noalloc": (C.c_int, [_H, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "trag_store_fill_synthetic": (C.c_int, [_H, C.c_uint64, C.c_uint64, C.c_uint64, _u32p]),
    "trag_shard_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "trag_shard_group_create": (C.c_int, [C.POINTER(ShardConfig), C.POINTER(C.c_uint8), C.POINTER(_H)]),
    "trag_shard_group_destroy": (None, [_H]),
    "trag_shard_search_device": (C.c_int, [_H, _H, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32,
                                           C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "trag_shard_fence": (C.c_int, [_H, C.c_void_p]),
    "trag_shard_search_batch": (C.c_int, [_H, _H, C.c_uint32, C.c_uint32, _fp, C.c_uint32, C.c_uint32, C.c_uint32,
                                          _u32p, _fp, _u32p]),
    "trag_shard_rccl_version": (C.c_int, [C.POINTER(C.c_int)]),
    "trag_shard_group_info": (C.c_int, [_H, _u32p, _u32p, C.POINTER(C.c_int32)]),
    "trag_sharded_create": (C.c_int, [C.POINTER(Config), C.c_uint32, C.POINTER(C.c_int32), C.c_uint32,
                                      C.POINTER(_H)]),
    "trag_sharded_destroy": (None, [_H]),
    "trag_sharded_insert": (C.c_int, [_H, _fp, C.c_uint64, C.c_uint32, _u32p]),
    "trag_sharded_overwrite": (C.c_int, [_H, C.c_uint32, _fp, C.c_uint32]),
    "trag_sharded_remove": (C.c_int, [_H, C.c_uint32]),
    "trag_sharded_len": (C.c_uint64, [_H]),
    "trag_sharded_rows": (C.c_uint64, [_H]),
    "trag_sharded_search": (C.c_int, [_H, _fp, C.c_uint32, C.c_uint32, _u32p, _fp, _u32p]),
    "trag_sharded_search_batch": (C.c_int, [_H, _fp, C.c_uint32, C.c_uint32, C.c_uint32, _u32p, _fp, _u32p]),
    "trag_sharded_fill_synthetic": (C.c_int, [_H, C.c_uint64, C.c_uint64, _u32p]),
    "trag_sharded_devices": (C.c_uint32, [_H]),
    "trag_sharded_shard": (_H, [_H, C.c_uint32]),
    "trag_merge_topk_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "trag_pack_topk_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                        C.c_void_p, C.c_void_p]),
    "trag_merge_packed_device": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p]),
    "trag_store_set_scan_mode": (C.c_int, [_H, C.c_uint32]),
    "trag_store_set_profiling": (C.c_int, [_H, C.c_int]),
    "trag_store_set_bf16_mirror": (C.c_int, [_H, C.c_int]),
    "trag_store_has_bf16_mirror": (C.c_int, [_H]),
    "trag_store_scan_time": (C.c_int, [_H, C.POINTER(C.c_double), _u32p]),
    "trag_store_stream_probe": (C.c_int, [_H, C.c_uint32, C.POINTER(C.c_double)]),
    "trag_store_read_rows": (C.c_int, [_H, C.c_uint64, C.c_uint64, _fp]),
    "trag_fuse": (C.c_int, [C.c_uint32, C.c_float, _u64p, _fp, C.c_uint32, _u64p, _fp, C.c_uint32,
                            _u64p, _fp, _u32p]),
    "trag_fuse_rrf": (C.c_int, [_u64p, C.c_uint32, _u64p, C.c_uint32, C.c_float, _u64p, _fp, _u32p]),
    "trag_fuse_rrf_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32,
                                       C.c_uint32, C.c_float, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p]),
    "trag_merge_topk": (C.c_int, [_u32p, _fp, _u32p, C.c_uint32, C.c_uint32, C.c_uint32, _u32p, _fp, _u32p]),
    "trag_store_last_stats": (C.c_int, [_H, C.POINTER(SearchStats)]),
    "trag_bm25_create": (C.c_int, [C.POINTER(BM25Config), C.POINTER(_H)]),
    "trag_bm25_destroy": (None, [_H]),
    "trag_bm25_add": (C.c_int, [_H, _u32p, _u64p, C.c_uint64, _u32p]),
    "trag_bm25_remove": (C.c_int, [_H, C.c_uint32]),
    "trag_bm25_len": (C.c_uint64, [_H]),
    "trag_bm25_search": (C.c_int, [_H, _u32p, C.c_uint32, C.c_uint32, _u32p, _fp, _u32p]),
    "trag_bm25_search_device": (C.c_int, [_H, _u32p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "trag_bm25_fill_synthetic": (C.c_int, [_H, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64,
                                           _u32p]),
    "trag_bm25_stats": (C.c_int, [_H, _u64p, _u64p]),
    "trag_mock_embed": (C.c_int, [C.c_char_p, C.c_uint64, C.c_uint32, C.c_int, _fp]),
    "trag_last_error": (C.c_char_p, []),
    "trag_version": (C.c_char_p, []),
}

_lib = None


def lib_path():
    return _LIB


def lib():
    """Load libtrag.so (built by `make -C trueno-rag_amd/csrc`). No fallback."""
    global _lib
    if _lib is None:
        # One process must hold one HIP runtime. PyTorch wheels bundle their own
        # libamdhip64 (same SONAME as /opt/rocm's): importing torch first makes
        # the dynamic loader bind libtrag.so to that already-loaded copy, so
        # torch device pointers and streams are valid inside this library.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        if not os.path.exists(_LIB):
            raise ImportError(f"{_LIB} is missing: build it with `make -C trueno-rag_amd/csrc` "
                              "(there is no CPU implementation to fall back to)")
        L = C.CDLL(_LIB)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def hip_runtimes_loaded():
    """Paths of every libamdhip64 mapped into this process (must be exactly one)."""
    seen = set()
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                seen.add(line.split()[-1])
    return sorted(seen)


def check(status, dim_expected=None, dim_actual=None):
    if status == OK:
        return
    msg = lib().trag_last_error().decode("utf-8", "replace")
    if status == DIM_MISMATCH:
        raise DimensionMismatch(dim_expected, dim_actual, msg)
    if status == INVALID_CONFIG:
        raise InvalidConfig(msg)
    raise TragError(status, msg)
