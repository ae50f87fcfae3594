"""Retrievers on the hot path (src/retrieve.rs of the reference): the
orchestration embed_query -> VectorStore.search -> sparse search -> fuse."""
from dataclasses import dataclass, field
from typing import List, Optional

from .fusion import FusionStrategy
from .index import Chunk, VectorStore


@dataclass
class RetrievalResult:
    """src/retrieve.rs:13-76"""
    chunk: Chunk
    dense_score: Optional[float] = None
    sparse_score: Optional[float] = None
    fused_score: Optional[float] = None
    rerank_score: Optional[float] = None

    def with_dense_score(self, s):
        self.dense_score = s
        return self

    def with_sparse_score(self, s):
        self.sparse_score = s
        return self

    def with_fused_score(self, s):
        self.fused_score = s
        return self

    def with_rerank_score(self, s):
        self.rerank_score = s
        return self

    def best_score(self) -> float:
        for s in (self.rerank_score, self.fused_score, self.dense_score, self.sparse_score):
            if s is not None:
                return s
        return 0.0


@dataclass
class HybridRetrieverConfig:
    """src/retrieve.rs:80-100"""
    candidates_per_source: int = 50
    fusion: FusionStrategy = field(default_factory=FusionStrategy.default)
    use_dense: bool = True
    use_sparse: bool = True


class HybridRetriever:
    """src/retrieve.rs:103-263. `sparse` is any object with add(chunk) and
    search(query, k) -> [(ChunkId, score)] (the SparseIndex trait,
    src/index.rs:8-28); `embedder` any object with embed_query(str) -> vector
    (the Embedder trait, src/embed.rs:55-89)."""

    def __init__(self, dense: VectorStore, sparse, embedder):
        self.dense = dense
        self.sparse = sparse
        self.embedder = embedder
        self.config = HybridRetrieverConfig()

    @classmethod
    def new(cls, dense, sparse, embedder):
        return cls(dense, sparse, embedder)

    def with_config(self, config: HybridRetrieverConfig):
        self.config = config
        return self

    def dense_store(self):
        return self.dense

    def sparse_index(self):
        return self.sparse

    def index(self, chunk: Chunk) -> None:
        self.sparse.add(chunk)          # src/retrieve.rs:158
        self.dense.insert(chunk)        # src/retrieve.rs:161

    def index_batch(self, chunks: List[Chunk]) -> None:
        for c in chunks:
            self.index(c)

    def retrieve(self, query: str, k: int) -> List[RetrievalResult]:
        """src/retrieve.rs:175-220"""
        candidates = self.config.candidates_per_source
        dense_results = []
        if self.config.use_dense:
            q = self.embedder.embed_query(query)
            dense_results = self.dense.search(q, candidates)
        sparse_results = self.sparse.search(query, candidates) if self.config.use_sparse else []
        fused = self.config.fusion.fuse(dense_results, sparse_results)
        dense_scores = dict(dense_results)
        sparse_scores = dict(sparse_results)
        results = []
        for chunk_id, fused_score in fused[:k]:
            chunk = self.dense.get(chunk_id)
            if chunk is None:          # ids unknown to the dense store are dropped (:205)
                continue
            r = RetrievalResult(chunk).with_fused_score(fused_score)
            if chunk_id in dense_scores:
                r = r.with_dense_score(dense_scores[chunk_id])
            if chunk_id in sparse_scores:
                r = r.with_sparse_score(sparse_scores[chunk_id])
            results.append(r)
        return results

    def retrieve_dense(self, query: str, k: int) -> List[RetrievalResult]:
        """src/retrieve.rs:223-235"""
        q = self.embedder.embed_query(query)
        out = []
        for chunk_id, score in self.dense.search(q, k):
            chunk = self.dense.get(chunk_id)
            if chunk is not None:
                out.append(RetrievalResult(chunk).with_dense_score(score))
        return out

    def retrieve_sparse(self, query: str, k: int) -> List[RetrievalResult]:
        """src/retrieve.rs:238-250"""
        out = []
        for chunk_id, score in self.sparse.search(query, k):
            chunk = self.dense.get(chunk_id)
            if chunk is not None:
                out.append(RetrievalResult(chunk).with_sparse_score(score))
        return out

    def len(self) -> int:
        return self.dense.len()

    def is_empty(self) -> bool:
        return self.dense.is_empty()


class DenseRetriever:
    """src/retrieve.rs:266-297"""

    def __init__(self, store: VectorStore, embedder):
        self.store = store
        self.embedder = embedder

    @classmethod
    def new(cls, store, embedder):
        return cls(store, embedder)

    def index(self, chunk: Chunk) -> None:
        self.store.insert(chunk)

    def retrieve(self, query: str, k: int) -> List[RetrievalResult]:
        q = self.embedder.embed_query(query)
        out = []
        for chunk_id, score in self.store.search(q, k):
            chunk = self.store.get(chunk_id)
            if chunk is not None:
                out.append(RetrievalResult(chunk).with_dense_score(score))
        return out


class SparseRetriever:
    """src/retrieve.rs:300-335: sparse-only retrieval. `index` is any SparseIndex (add(chunk),
    search(query, k) -> [(ChunkId, score)]); by default the GPU-resident BM25Index. The chunks are kept
    here, keyed by id, as the reference keeps them."""

    def __init__(self, index=None):
        if index is None:
            from .sparse import BM25Index
            index = BM25Index.new()
        self._index = index
        self.chunks = {}

    @classmethod
    def new(cls):
        return cls()

    @classmethod
    def with_index(cls, index):
        return cls(index)

    def index(self, chunk: Chunk) -> None:
        self._index.add(chunk)                 # src/retrieve.rs:317
        self.chunks[chunk.id] = chunk          # :318

    def retrieve(self, query: str, k: int) -> List[RetrievalResult]:
        out = []
        for chunk_id, score in self._index.search(query, k):
            chunk = self.chunks.get(chunk_id)  # ids without a chunk are dropped (:329-331)
            if chunk is not None:
                out.append(RetrievalResult(chunk).with_sparse_score(score))
        return out
