"""VectorStore over the HIP library: the reference's method surface
(src/index.rs:282-437) with the same names, argument meaning and errors."""
import ctypes as C
import enum
import uuid
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import _abi


class DistanceMetric(enum.IntEnum):
    """src/index.rs:310-319"""
    Cosine = 0
    Euclidean = 1
    DotProduct = 2


class ScanMode(enum.IntEnum):
    """trag_scan_mode of include/trag.h (no counterpart in the reference: results are the
    same in every mode, only the route to them differs)."""
    Auto = 0
    Exact = 1
    Stream = 2
    Mfma = 3
    Bf16Prefilter = 4
    Mirror = 5


@dataclass
class VectorStoreConfig:
    """src/index.rs:283-307. The hnsw_* fields exist for API parity; like the
    reference, nothing reads them (search is always exhaustive)."""
    dimension: int = 384
    metric: DistanceMetric = DistanceMetric.Cosine
    hnsw_m: int = 16
    hnsw_ef_construction: int = 100
    hnsw_ef_search: int = 50
    # not in the reference: device placement and scan mode of this implementation
    device: int = -1
    scan_mode: int = 0
    reserve_rows: int = 0


class ChunkId:
    """src/chunk.rs:10-16: a random v4 UUID; from_u128 for deterministic tests."""
    __slots__ = ("value",)

    def __init__(self, value: Optional[uuid.UUID] = None):
        self.value = value if value is not None else uuid.uuid4()

    @classmethod
    def from_u128(cls, n: int) -> "ChunkId":
        return cls(uuid.UUID(int=n))

    def __eq__(self, other):
        return isinstance(other, ChunkId) and self.value == other.value

    def __hash__(self):
        return hash(self.value)

    def __repr__(self):
        return f"ChunkId({self.value})"


@dataclass
class Chunk:
    """The fields of src/chunk.rs:50-99 that the retrieval path touches."""
    content: str = ""
    embedding: Optional[List[float]] = None
    id: ChunkId = field(default_factory=ChunkId)
    document_id: Optional[uuid.UUID] = None
    start_offset: int = 0
    end_offset: int = 0

    def with_embedding(self, embedding) -> "Chunk":
        self.embedding = list(embedding)
        return self


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class VectorStore:
    """Dense vector store resident in HBM. Row indices live on this side:
    id -> row and row -> id tables plus the Chunk cache (`chunks` in the
    reference, src/index.rs:329)."""

    def __init__(self, config: Optional[VectorStoreConfig] = None):
        self._config = config or VectorStoreConfig()
        cfg = _abi.Config(self._config.dimension, int(self._config.metric), self._config.device,
                          self._config.scan_mode, self._config.reserve_rows)
        self._h = C.c_void_p()
        _abi.check(_abi.lib().trag_store_create(C.byref(cfg), C.byref(self._h)))
        self._row_of: Dict[ChunkId, int] = {}
        self._id_of: Dict[int, ChunkId] = {}
        self._chunks: Dict[ChunkId, Chunk] = {}

    @classmethod
    def new(cls, config: VectorStoreConfig) -> "VectorStore":
        return cls(config)

    @classmethod
    def with_dimension(cls, dimension: int) -> "VectorStore":
        return cls(VectorStoreConfig(dimension=dimension))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            _abi.lib().trag_store_destroy(h)
            self._h = None

    def close(self):
        self.__del__()

    def clone(self) -> "VectorStore":
        """`#[derive(Clone)]`, src/index.rs:322: an independent store with the same rows (copied
        inside HBM), ids and chunks."""
        other = object.__new__(type(self))
        other._config = self._config
        other._h = C.c_void_p()
        _abi.check(_abi.lib().trag_store_clone(self._h, C.byref(other._h)))
        other._row_of = dict(self._row_of)
        other._id_of = dict(self._id_of)
        other._chunks = dict(self._chunks)
        return other

    __copy__ = clone

    def __repr__(self):   # `#[derive(Debug)]`
        return f"VectorStore(config={self._config!r}, len={self.len()})"

    def config(self) -> VectorStoreConfig:
        return self._config

    # -- src/index.rs:359-375
    def insert(self, chunk: Chunk) -> None:
        if chunk.embedding is None:
            raise _abi.InvalidConfig("chunk must have embedding")
        emb = np.ascontiguousarray(chunk.embedding, dtype=np.float32)
        dim = self._config.dimension
        if emb.size != dim:
            raise _abi.DimensionMismatch(dim, int(emb.size))
        L = _abi.lib()
        row = self._row_of.get(chunk.id)
        if row is not None:  # HashMap::insert overwrites (src/index.rs:372)
            _abi.check(L.trag_store_overwrite(self._h, row, _fptr(emb), emb.size), dim, emb.size)
        else:
            first = C.c_uint32()
            _abi.check(L.trag_store_insert(self._h, _fptr(emb), 1, emb.size, C.byref(first)), dim, emb.size)
            self._row_of[chunk.id] = first.value
            self._id_of[first.value] = chunk.id
        self._chunks[chunk.id] = chunk

    # -- src/index.rs:378-383 (stops at the first failing chunk, earlier ones stay)
    def insert_batch(self, chunks: List[Chunk]) -> None:
        # fast path: all new, all well-formed -> one H2D copy
        dim = self._config.dimension
        ok = all(c.embedding is not None and len(c.embedding) == dim and c.id not in self._row_of
                 for c in chunks) and len({c.id for c in chunks}) == len(chunks)
        if ok and chunks:
            mat = np.ascontiguousarray([c.embedding for c in chunks], dtype=np.float32)
            first = C.c_uint32()
            _abi.check(_abi.lib().trag_store_insert(self._h, _fptr(mat), len(chunks), dim, C.byref(first)),
                       dim, dim)
            for i, c in enumerate(chunks):
                self._row_of[c.id] = first.value + i
                self._id_of[first.value + i] = c.id
                self._chunks[c.id] = c
            return
        for c in chunks:
            self.insert(c)

    def insert_rows(self, rows: np.ndarray) -> int:
        """Bulk append of an (n, dim) float32 matrix without Chunk objects; rows get
        fresh ChunkIds lazily. Returns the first row index. Not in the reference."""
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        if rows.ndim != 2:
            raise _abi.InvalidConfig("rows must be a 2-D array")
        first = C.c_uint32()
        _abi.check(_abi.lib().trag_store_insert(self._h, _fptr(rows), rows.shape[0], rows.shape[1],
                                                C.byref(first)), self._config.dimension, rows.shape[1])
        return first.value

    def fill_synthetic(self, n: int, seed: int, first_global_row: Optional[int] = None) -> int:
        """Append n generator rows; generator row index starts at first_global_row
        (default: the store's current row count)."""
        first = C.c_uint32()
        g0 = self.rows() if first_global_row is None else first_global_row
        _abi.check(_abi.lib().trag_store_fill_synthetic(self._h, n, seed, g0, C.byref(first)))
        return first.value

    def rows(self) -> int:
        return int(_abi.lib().trag_store_rows(self._h))

    def set_scan_mode(self, mode: int) -> None:
        _abi.check(_abi.lib().trag_store_set_scan_mode(self._h, mode))

    def set_profiling(self, enabled) -> None:
        """False/0 off, True/1 every scan launch, n > 1 every n-th scan launch."""
        _abi.check(_abi.lib().trag_store_set_profiling(self._h, int(enabled)))

    def set_bf16_mirror(self, enabled: bool) -> None:
        """Build (or free) the bf16 copy of the corpus that AUTO / ScanMode.Mirror read."""
        _abi.check(_abi.lib().trag_store_set_bf16_mirror(self._h, 1 if enabled else 0))

    def has_bf16_mirror(self) -> bool:
        return bool(_abi.lib().trag_store_has_bf16_mirror(self._h))

    def scan_time(self):
        """(total milliseconds, launches) of the scan kernels since profiling was enabled."""
        ms, n = C.c_double(), C.c_uint32()
        _abi.check(_abi.lib().trag_store_scan_time(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def stream_probe(self, repeats: int = 3) -> float:
        """GB/s of a plain contiguous read of the store's rows (a reference point for rooflines)."""
        g = C.c_double()
        _abi.check(_abi.lib().trag_store_stream_probe(self._h, repeats, C.byref(g)))
        return g.value

    def search_device(self, d_queries_ptr: int, n_queries: int, k: int, d_rows_ptr: int, d_scores_ptr: int,
                      d_n_ptr: int, d_status_ptr: int = 0, stream: int = 0) -> None:
        """Asynchronous search on device-resident queries (raw device addresses)."""
        _abi.check(_abi.lib().trag_store_search_device(
            self._h, C.c_void_p(d_queries_ptr), n_queries, self._config.dimension, k, C.c_void_p(d_rows_ptr),
            C.c_void_p(d_scores_ptr), C.c_void_p(d_n_ptr), C.c_void_p(d_status_ptr) if d_status_ptr else None,
            C.c_void_p(stream) if stream else None), self._config.dimension, self._config.dimension)

    def set_pipelined(self, enabled) -> None:
        """Single-query device-entry searches overlap their select with the next query's scan; outputs
        are valid on a stream after fence(stream). 1 / True: the select runs on the handle's side
        stream; 2: it rides as one more workgroup in the launch of the next query's scan."""
        _abi.check(_abi.lib().trag_store_set_pipelined(self._h, int(enabled)))

    def fence(self, stream: int = 0) -> None:
        _abi.check(_abi.lib().trag_store_fence(self._h, C.c_void_p(stream) if stream else None))

    def search_device_noalloc(self, d_queries_ptr: int, n_queries: int, k: int, d_rows_ptr: int,
                              d_scores_ptr: int, d_n_ptr: int, d_status_ptr: int = 0, stream: int = 0) -> None:
        """search_device for callers that cannot allocate: exact single-query scans whose scratch is
        the library's fixed device array (shared by every store on the GPU)."""
        _abi.check(_abi.lib().trag_store_search_device_noalloc(
            self._h, C.c_void_p(d_queries_ptr), n_queries, self._config.dimension, k, C.c_void_p(d_rows_ptr),
            C.c_void_p(d_scores_ptr), C.c_void_p(d_n_ptr), C.c_void_p(d_status_ptr) if d_status_ptr else None,
            C.c_void_p(stream) if stream else None), self._config.dimension, self._config.dimension)

    def read_rows(self, row0: int, n: int) -> np.ndarray:
        out = np.empty((n, self._config.dimension), dtype=np.float32)
        _abi.check(_abi.lib().trag_store_read_rows(self._h, row0, n, _fptr(out)))
        return out

    def _id_for_row(self, row: int) -> ChunkId:
        cid = self._id_of.get(row)
        if cid is None:  # bulk-inserted row: mint its id on first sight
            cid = ChunkId()
            self._id_of[row] = cid
            self._row_of[cid] = row
        return cid

    # -- src/index.rs:386-412
    def search_rows(self, query_vector, k: int) -> Tuple[np.ndarray, np.ndarray]:
        """(row indices, scores), best first."""
        q = np.ascontiguousarray(query_vector, dtype=np.float32).reshape(-1)
        dim = self._config.dimension
        if q.size != dim:
            raise _abi.DimensionMismatch(dim, int(q.size))
        k = min(int(k), self.len())   # the reference truncates: any k is legal, at most len() rows come back
        rows = np.empty(max(k, 1), dtype=np.uint32)
        scores = np.empty(max(k, 1), dtype=np.float32)
        n = C.c_uint32()
        _abi.check(_abi.lib().trag_store_search(self._h, _fptr(q), q.size, k,
                                                rows.ctypes.data_as(C.POINTER(C.c_uint32)), _fptr(scores),
                                                C.byref(n)), dim, q.size)
        return rows[:n.value].copy(), scores[:n.value].copy()

    def search_batch_rows(self, queries, k: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        Q = np.ascontiguousarray(queries, dtype=np.float32)
        if Q.ndim != 2:
            raise _abi.InvalidConfig("queries must be a 2-D array")
        dim = self._config.dimension
        if Q.shape[1] != dim:
            raise _abi.DimensionMismatch(dim, int(Q.shape[1]))
        B = Q.shape[0]
        k = min(int(k), self.len())
        rows = np.zeros((B, max(k, 1)), dtype=np.uint32)
        scores = np.zeros((B, max(k, 1)), dtype=np.float32)
        n = np.zeros(B, dtype=np.uint32)
        _abi.check(_abi.lib().trag_store_search_batch(self._h, _fptr(Q), B, dim, k,
                                                      rows.ctypes.data_as(C.POINTER(C.c_uint32)), _fptr(scores),
                                                      n.ctypes.data_as(C.POINTER(C.c_uint32))), dim, dim)
        return rows[:, :k], scores[:, :k], n

    def search(self, query_vector, k: int) -> List[Tuple[ChunkId, float]]:
        rows, scores = self.search_rows(query_vector, k)
        return [(self._id_for_row(int(r)), float(s)) for r, s in zip(rows, scores)]

    # -- src/index.rs:416-418
    def get(self, chunk_id: ChunkId) -> Optional[Chunk]:
        return self._chunks.get(chunk_id)

    # -- src/index.rs:421-424
    def remove(self, chunk_id: ChunkId) -> Optional[Chunk]:
        row = self._row_of.pop(chunk_id, None)
        if row is None:
            return None
        _abi.check(_abi.lib().trag_store_remove(self._h, row))
        self._id_of.pop(row, None)
        return self._chunks.pop(chunk_id, None)

    def remove_row(self, row: int) -> None:
        _abi.check(_abi.lib().trag_store_remove(self._h, row))
        cid = self._id_of.pop(row, None)
        if cid is not None:
            self._row_of.pop(cid, None)
            self._chunks.pop(cid, None)

    # -- src/index.rs:428-436
    def len(self) -> int:
        return int(_abi.lib().trag_store_len(self._h))

    def __len__(self):
        return self.len()

    def is_empty(self) -> bool:
        return self.len() == 0

    def last_stats(self) -> dict:
        st = _abi.SearchStats()
        _abi.check(_abi.lib().trag_store_last_stats(self._h, C.byref(st)))
        return {"mode_used": st.mode_used, "candidates": st.candidates,
                "refine_levels": st.refine_levels, "fallbacks": st.fallbacks,
                "keys_kept_from": st.keys_kept_from}

    @property
    def handle(self):
        return self._h
