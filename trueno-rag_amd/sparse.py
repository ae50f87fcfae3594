"""BM25Index over the HIP library: the reference's sparse index
(src/index.rs:7-280) with the same names, argument meaning and results.

The tokenizer and the term dictionary stay on the host, as they would in the
reference's Rust; documents and queries cross the C ABI as term-id sequences.
Scoring and top-k run on the GPU (trag_bm25_* in include/trag.h). There is no
CPU scoring path here.
"""
import ctypes as C
from typing import Dict, Iterable, List, Optional, Set, Tuple

import numpy as np

from . import _abi
from .index import Chunk, ChunkId

# src/index.rs:93-108
DEFAULT_STOPWORDS = frozenset("""
a an the is are was were be been being have has had do does did will would could should may might must
shall can need dare ought used to of in for on with at by from as into through during before after above
below between under again further then once here there when where why how all each few more most other some
such no nor not only own same so than too very just and but if or because until while this that these
those it its
""".split())


def _u32ptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


class BM25Index:
    """src/index.rs:31-51 (fields) and :59-173, :175-280 (methods)."""

    def __init__(self, k1: float = 1.2, b: float = 0.75, device: int = -1):
        self.k1 = float(k1)
        self.b = float(b)
        self.lowercase = True
        self.stopwords: Set[str] = set(DEFAULT_STOPWORDS)
        self._terms: Dict[str, int] = {}        # term -> id (the keys of `inverted_index`)
        self._doc_of: Dict[ChunkId, List[int]] = {}  # id -> documents (a re-added id has several)
        self._id_of: List[Optional[ChunkId]] = []
        self._h = _abi._H()
        cfg = _abi.BM25Config(self.k1, self.b, device)
        _abi.check(_abi.lib().trag_bm25_create(C.byref(cfg), C.byref(self._h)))

    # -- constructors of the reference -------------------------------------
    @classmethod
    def new(cls) -> "BM25Index":
        return cls()

    @classmethod
    def with_params(cls, k1: float, b: float) -> "BM25Index":
        return cls(k1, b)

    def with_stopwords(self, stopwords: Iterable[str]) -> "BM25Index":
        self.stopwords = set(stopwords)
        return self

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _abi.lib().trag_bm25_destroy(self._h)
            self._h = _abi._H()

    # -- tokenizer (src/index.rs:111-124) -----------------------------------
    def tokenize(self, text: str) -> List[str]:
        """Split at every non-alphanumeric character, lowercase, drop stop words,
        drop tokens shorter than 2 BYTES. str.isalnum() stands in for Rust's
        char::is_alphanumeric: identical on ASCII; they differ on a few exotic
        code points (Rust counts Other_Alphabetic marks, Python does not)."""
        out, cur = [], []
        for ch in text:
            if ch.isalnum():
                cur.append(ch)
            elif cur:
                out.append("".join(cur))
                cur = []
        if cur:
            out.append("".join(cur))
        if self.lowercase:
            out = [s.lower() for s in out]
        return [s for s in out if s not in self.stopwords and len(s.encode("utf-8")) >= 2]

    def _ids(self, tokens: List[str], create: bool) -> np.ndarray:
        ids = []
        for t in tokens:
            i = self._terms.get(t)
            if i is None:
                if not create:
                    i = 0xFFFFFFF  # below the ABI's id limit, never assigned: matches nothing
                else:
                    i = len(self._terms)
                    self._terms[t] = i
            ids.append(i)
        return np.asarray(ids, dtype=np.uint32)

    # -- SparseIndex (src/index.rs:175-280) ---------------------------------
    def add(self, chunk: Chunk) -> None:
        self.add_batch([chunk])

    def add_batch(self, chunks: List[Chunk]) -> None:
        if not chunks:
            return
        # Re-adding a ChunkId REPLACES its document. (The reference keeps both postings, counts the
        # chunk twice in doc_count and df while doc_lengths holds one entry, and scores it once from
        # the first posting's tf and the newer length, src/index.rs:176-204,127-133; that
        # bookkeeping is not reproduced: a chunk is one document here, so a search never returns an
        # id twice and fuse never counts it twice.)
        if len({c.id for c in chunks}) != len(chunks):   # the same id twice in one batch: one by one
            for c in chunks:
                self.add_batch([c])
            return
        for c in chunks:
            if c.id in self._doc_of:
                self.remove(c.id)
        seqs = [self._ids(self.tokenize(c.content), True) for c in chunks]
        off = np.zeros(len(seqs) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(s) for s in seqs])
        terms = np.concatenate(seqs).astype(np.uint32) if off[-1] else np.zeros(1, dtype=np.uint32)
        first = C.c_uint32(0)
        _abi.check(_abi.lib().trag_bm25_add(self._h, _u32ptr(terms), off.ctypes.data_as(C.POINTER(C.c_uint64)),
                                            len(seqs), C.byref(first)))
        for i, c in enumerate(chunks):
            self._doc_of.setdefault(c.id, []).append(first.value + i)
            self._id_of.append(c.id)

    def add_token_ids(self, terms: np.ndarray, doc_offsets: np.ndarray) -> int:
        """Bulk path for already tokenised documents (ids from the caller's own
        dictionary); returns the first document index."""
        terms = np.ascontiguousarray(terms, dtype=np.uint32)
        off = np.ascontiguousarray(doc_offsets, dtype=np.uint64)
        first = C.c_uint32(0)
        _abi.check(_abi.lib().trag_bm25_add(self._h, _u32ptr(terms if terms.size else np.zeros(1, np.uint32)),
                                            off.ctypes.data_as(C.POINTER(C.c_uint64)), len(off) - 1,
                                            C.byref(first)))
        self._id_of.extend([None] * (len(off) - 1))
        return first.value

    def fill_synthetic(self, n_docs: int, vocab: int, tokens_per_doc: int, seed: int,
                       first_global_doc: int = 0) -> int:
        first = C.c_uint32(0)
        _abi.check(_abi.lib().trag_bm25_fill_synthetic(self._h, n_docs, vocab, tokens_per_doc, seed,
                                                       first_global_doc, C.byref(first)))
        self._id_of.extend([None] * n_docs)
        return first.value

    def search_ids(self, query_terms, k: int) -> Tuple[np.ndarray, np.ndarray]:
        """Search with a tokenised query; returns (document indices, scores)."""
        q = np.ascontiguousarray(query_terms, dtype=np.uint32)
        k = min(int(k), self.len())   # the reference truncates: any k is legal
        docs = np.zeros(max(k, 1), dtype=np.uint32)
        scores = np.zeros(max(k, 1), dtype=np.float32)
        n = C.c_uint32(0)
        _abi.check(_abi.lib().trag_bm25_search(self._h, _u32ptr(q if q.size else np.zeros(1, np.uint32)), q.size,
                                               k, _u32ptr(docs), scores.ctypes.data_as(C.POINTER(C.c_float)),
                                               C.byref(n)))
        return docs[:n.value].copy(), scores[:n.value].copy()

    def search_ids_device(self, query_terms, k: int, d_docs_ptr: int, d_scores_ptr: int, d_n_ptr: int,
                          d_status_ptr: int, stream: int = 0) -> None:
        """search_ids with device-resident outputs (k entries, one count, one status word), enqueued
        on `stream` without synchronising; status 3 asks the caller to repeat the query through
        search_ids (trag_bm25_search_device). The query terms are host memory."""
        q = np.ascontiguousarray(query_terms, dtype=np.uint32)
        _abi.check(_abi.lib().trag_bm25_search_device(self._h, _u32ptr(q if q.size else np.zeros(1, np.uint32)),
                                                      q.size, k, d_docs_ptr, d_scores_ptr, d_n_ptr, d_status_ptr,
                                                      stream or None))

    def search(self, query: str, k: int) -> List[Tuple[ChunkId, float]]:
        terms = self.tokenize(query)
        if not terms:
            return []                     # src/index.rs:214-216
        docs, scores = self.search_ids(self._ids(terms, False), k)
        return [(self._id_of[int(d)], float(s)) for d, s in zip(docs, scores)]

    def remove(self, chunk_id: ChunkId) -> None:
        for d in self._doc_of.pop(chunk_id, []):
            _abi.check(_abi.lib().trag_bm25_remove(self._h, d))
            self._id_of[d] = None

    def remove_doc(self, doc: int) -> None:
        _abi.check(_abi.lib().trag_bm25_remove(self._h, doc))

    def len(self) -> int:
        return int(_abi.lib().trag_bm25_len(self._h))

    def __len__(self):
        return self.len()

    def is_empty(self) -> bool:
        return self.len() == 0

    def stats(self) -> dict:
        p, t = C.c_uint64(0), C.c_uint64(0)
        _abi.check(_abi.lib().trag_bm25_stats(self._h, C.byref(p), C.byref(t)))
        return {"postings": p.value, "terms": t.value}

    @property
    def handle(self):
        return self._h
