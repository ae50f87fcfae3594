"""Host-side mirror (Python, over the C ABI in include/trag.h) of the part of
trueno-rag's API that sits on the dense-retrieval hot path.

The directory name contains a hyphen, so load the package with
`importlib` (see `load_package()` in the repository's `__graft_entry__.py`)
under the module name `trueno_rag_amd`.

There is no CPU fallback: every search goes through libtrag.so and its HIP
kernels, and loading fails loudly when the library is missing.
"""
from ._abi import lib, lib_path, TragError, DimensionMismatch, InvalidConfig  # noqa: F401
from .index import VectorStore, VectorStoreConfig, DistanceMetric, ScanMode, Chunk, ChunkId  # noqa: F401
from .sparse import BM25Index  # noqa: F401
from .embed import MockEmbedder, EmbeddingConfig, EmptyDocument, QueryError  # noqa: F401
from .persist import PersistedIndex, PersistedChunk, IndexServer  # noqa: F401
from .fusion import FusionStrategy  # noqa: F401
from .rerank import LexicalReranker, NoOpReranker  # noqa: F401
from .retrieve import (DenseRetriever, HybridRetriever, HybridRetrieverConfig,  # noqa: F401
                       RetrievalResult, SparseRetriever)

__all__ = ["lib", "lib_path", "TragError", "DimensionMismatch", "InvalidConfig", "VectorStore",
           "VectorStoreConfig", "DistanceMetric", "ScanMode", "Chunk", "ChunkId", "BM25Index", "MockEmbedder", "EmbeddingConfig", "EmptyDocument", "QueryError", "PersistedIndex", "PersistedChunk", "IndexServer", "FusionStrategy",
           "LexicalReranker", "NoOpReranker", "DenseRetriever", "HybridRetriever", "HybridRetrieverConfig", "RetrievalResult", "SparseRetriever"]
