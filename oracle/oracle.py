"""ctypes wrapper around oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module. It is the checker, never the thing measured or shipped.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

COSINE, EUCLIDEAN, DOT = 0, 1, 2
FUSE_RRF, FUSE_LINEAR, FUSE_CONVEX, FUSE_DBSF, FUSE_UNION, FUSE_INTERSECTION = range(6)


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("trag_oracle.c", "trag_oracle_bm25.c")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(map(os.path.getmtime, srcs)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        fp, u8p, u32p, u64p = (C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_uint32),
                               C.POINTER(C.c_uint64))
        L.oracle_dot.restype = C.c_float
        L.oracle_dot.argtypes = [fp, fp, C.c_uint32]
        L.oracle_cosine.restype = C.c_float
        L.oracle_cosine.argtypes = [fp, fp, C.c_uint32]
        L.oracle_euclidean.restype = C.c_float
        L.oracle_euclidean.argtypes = [fp, fp, C.c_uint32]
        L.oracle_score_all.restype = None
        L.oracle_score_all.argtypes = [fp, C.c_uint64, C.c_uint32, C.c_int, fp, fp]
        L.oracle_search.restype = C.c_uint32
        L.oracle_search.argtypes = [fp, u8p, C.c_uint64, C.c_uint32, C.c_int, fp, C.c_uint32, u32p, fp]
        L.oracle_search_faithful.restype = C.c_uint32
        L.oracle_search_faithful.argtypes = [C.POINTER(fp), u32p, C.c_uint64, C.c_uint32, C.c_int, fp,
                                             C.c_uint32, u32p, fp]
        L.oracle_rrf.restype = C.c_uint32
        L.oracle_rrf.argtypes = [u64p, C.c_uint32, u64p, C.c_uint32, C.c_float, u64p, fp]
        L.oracle_fuse.restype = C.c_uint32
        L.oracle_fuse.argtypes = [C.c_int, C.c_float, u64p, fp, C.c_uint32, u64p, fp, C.c_uint32, u64p, fp]
        L.oracle_synth_component.restype = C.c_float
        L.oracle_synth_component.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
        L.oracle_synth_rows.restype = None
        L.oracle_synth_rows.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, fp]
        L.oracle_num_threads.restype = C.c_int
        L.oracle_bench_faithful.restype = C.c_double
        L.oracle_bench_faithful.argtypes = [fp, C.c_uint64, C.c_uint32, C.c_int, fp, C.c_uint32, C.c_uint32, u32p, fp]
        L.oracle_bm25_search.restype = C.c_uint32
        L.oracle_bm25_search.argtypes = [u32p, u64p, C.c_uint64, u8p, C.c_float, C.c_float, u32p, C.c_uint32,
                                         C.c_uint32, u32p, fp]
        L.oracle_bm25_synth_docs.restype = None
        L.oracle_bm25_synth_docs.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, u32p]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def dot(a, b):
    a, b = _f32(a), _f32(b)
    return float(lib().oracle_dot(_p(a, C.c_float), _p(b, C.c_float), a.size))


def cosine(a, b):
    a, b = _f32(a), _f32(b)
    return float(lib().oracle_cosine(_p(a, C.c_float), _p(b, C.c_float), a.size))


def euclidean(a, b):
    a, b = _f32(a), _f32(b)
    return float(lib().oracle_euclidean(_p(a, C.c_float), _p(b, C.c_float), a.size))


def score_all(corpus, q, metric=COSINE):
    corpus, q = _f32(corpus), _f32(q)
    n, d = corpus.shape
    out = np.empty(n, dtype=np.float32)
    lib().oracle_score_all(_p(corpus, C.c_float), n, d, metric, _p(q, C.c_float), _p(out, C.c_float))
    return out


def search(corpus, q, k, metric=COSINE, dead=None):
    """Returns (rows uint32[m], scores float32[m]), m = min(k, live rows)."""
    corpus, q = _f32(corpus), _f32(q)
    n, d = corpus.shape if corpus.ndim == 2 else (0, q.size)
    rows = np.empty(max(k, 1), dtype=np.uint32)
    scores = np.empty(max(k, 1), dtype=np.float32)
    dp = None
    if dead is not None:
        dead = np.ascontiguousarray(dead, dtype=np.uint8)
        dp = _p(dead, C.c_uint8)
    m = lib().oracle_search(_p(corpus, C.c_float), dp, n, d, metric, _p(q, C.c_float), k,
                            _p(rows, C.c_uint32), _p(scores, C.c_float))
    return rows[:m].copy(), scores[:m].copy()


def search_faithful(corpus, q, k, metric=COSINE, order=None):
    """Reference-shaped single-thread search: per-row heap vectors, full stable sort."""
    corpus, q = _f32(corpus), _f32(q)
    n, d = corpus.shape
    rows_store = [np.array(corpus[i], dtype=np.float32) for i in range(n)]  # separate allocations
    ptrs = (C.POINTER(C.c_float) * n)(*[_p(r, C.c_float) for r in rows_store])
    rows = np.empty(max(k, 1), dtype=np.uint32)
    scores = np.empty(max(k, 1), dtype=np.float32)
    op = None
    if order is not None:
        order = np.ascontiguousarray(order, dtype=np.uint32)
        op = _p(order, C.c_uint32)
    m = lib().oracle_search_faithful(ptrs, op, n, d, metric, _p(q, C.c_float), k,
                                     _p(rows, C.c_uint32), _p(scores, C.c_float))
    return rows[:m].copy(), scores[:m].copy()


def rrf(dense_ids, sparse_ids, k=60.0):
    di = np.ascontiguousarray(dense_ids, dtype=np.uint64)
    si = np.ascontiguousarray(sparse_ids, dtype=np.uint64)
    cap = max(di.size + si.size, 1)
    out_ids = np.empty(cap, dtype=np.uint64)
    out_scores = np.empty(cap, dtype=np.float32)
    n = lib().oracle_rrf(_p(di, C.c_uint64), di.size, _p(si, C.c_uint64), si.size, k,
                         _p(out_ids, C.c_uint64), _p(out_scores, C.c_float))
    return out_ids[:n].copy(), out_scores[:n].copy()


def fuse(strategy, param, dense_ids, dense_scores, sparse_ids, sparse_scores):
    di = np.ascontiguousarray(dense_ids, dtype=np.uint64)
    si = np.ascontiguousarray(sparse_ids, dtype=np.uint64)
    ds, ss = _f32(dense_scores), _f32(sparse_scores)
    cap = max(di.size + si.size, 1)
    out_ids = np.empty(cap, dtype=np.uint64)
    out_scores = np.empty(cap, dtype=np.float32)
    n = lib().oracle_fuse(strategy, param, _p(di, C.c_uint64), _p(ds, C.c_float), di.size,
                          _p(si, C.c_uint64), _p(ss, C.c_float), si.size,
                          _p(out_ids, C.c_uint64), _p(out_scores, C.c_float))
    return out_ids[:n].copy(), out_scores[:n].copy()


def synth_rows(seed, row0, n, d, normalize=True):
    out = np.empty((n, d), dtype=np.float32)
    lib().oracle_synth_rows(seed, row0, n, d, 1 if normalize else 0, _p(out, C.c_float))
    return out


def bench_faithful(corpus, queries, k, metric=COSINE):
    """Median seconds per query of the reference-shaped single-thread search, plus
    the last query's (rows, scores)."""
    corpus, queries = _f32(corpus), _f32(queries)
    n, d = corpus.shape
    rows = np.empty(max(k, 1), dtype=np.uint32)
    scores = np.empty(max(k, 1), dtype=np.float32)
    sec = lib().oracle_bench_faithful(_p(corpus, C.c_float), n, d, metric, _p(queries, C.c_float),
                                      queries.shape[0], k, _p(rows, C.c_uint32), _p(scores, C.c_float))
    m = min(k, n)
    return float(sec), rows[:m].copy(), scores[:m].copy()


def num_threads():
    return int(lib().oracle_num_threads())


def bm25_search(terms, doc_offsets, q_terms, k, k1=1.2, b=0.75, removed=None):
    """BM25 over tokenised documents (src/index.rs:212-243 of the reference):
    returns (document indices, scores), best first, ties by ascending document."""
    terms = np.ascontiguousarray(terms, dtype=np.uint32)
    off = np.ascontiguousarray(doc_offsets, dtype=np.uint64)
    q = np.ascontiguousarray(q_terms, dtype=np.uint32)
    n_docs = len(off) - 1
    rm = None
    if removed is not None:
        rm = np.ascontiguousarray(removed, dtype=np.uint8)
        assert rm.shape == (n_docs,)
    docs = np.empty(max(k, 1), dtype=np.uint32)
    scores = np.empty(max(k, 1), dtype=np.float32)
    tp = _p(terms, C.c_uint32) if terms.size else _p(np.zeros(1, np.uint32), C.c_uint32)
    qp = _p(q, C.c_uint32) if q.size else _p(np.zeros(1, np.uint32), C.c_uint32)
    n = lib().oracle_bm25_search(tp, _p(off, C.c_uint64), n_docs, _p(rm, C.c_uint8) if rm is not None else None,
                                 k1, b, qp, q.size, k, _p(docs, C.c_uint32), _p(scores, C.c_float))
    return docs[:n].copy(), scores[:n].copy()


def bm25_synth_docs(seed, doc0, n_docs, vocab, tokens_per_doc):
    """Zipf(1.0) token ids, shape (n_docs, tokens_per_doc)."""
    out = np.empty((n_docs, tokens_per_doc), dtype=np.uint32)
    lib().oracle_bm25_synth_docs(seed, doc0, n_docs, vocab, tokens_per_doc, _p(out, C.c_uint32))
    return out
