"""Embedder surface of the reference that feeds the retrieval path
(src/embed.rs:26-197): EmbeddingConfig and MockEmbedder, the deterministic
hash embedder its tests and benches use as the query / corpus generator."""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import _abi


class EmptyDocument(ValueError):
    """Error::EmptyDocument(String), src/embed.rs:159-161."""


class QueryError(ValueError):
    """Error::Query(String), src/embed.rs:185-187."""


@dataclass
class EmbeddingConfig:
    """src/embed.rs:26-50 (the fields MockEmbedder reads)."""
    normalize: bool = True
    query_prefix: Optional[str] = None
    document_prefix: Optional[str] = None


@dataclass
class MockEmbedder:
    """src/embed.rs:91-197."""
    dimension_: int
    model_id_: str = "mock-embedder"
    config: EmbeddingConfig = field(default_factory=EmbeddingConfig)

    @classmethod
    def new(cls, dimension: int) -> "MockEmbedder":
        return cls(dimension)

    def with_model_id(self, model_id: str) -> "MockEmbedder":
        self.model_id_ = model_id
        return self

    def with_config(self, config: EmbeddingConfig) -> "MockEmbedder":
        self.config = config
        return self

    def _hash_to_vector(self, text: str) -> np.ndarray:
        raw = text.encode("utf-8")
        out = np.empty(self.dimension_, dtype=np.float32)
        _abi.check(_abi.lib().trag_mock_embed(raw, len(raw), self.dimension_, 1 if self.config.normalize else 0,
                                              out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def embed(self, text: str) -> np.ndarray:
        if not text:
            raise EmptyDocument("empty text for embedding")
        return self._hash_to_vector((self.config.document_prefix or "") + text)

    def embed_batch(self, texts: List[str]) -> List[np.ndarray]:
        return [self.embed(t) for t in texts]

    def embed_document(self, document: str) -> np.ndarray:
        return self.embed(document)

    def embed_query(self, query: str) -> np.ndarray:
        if not query:
            raise QueryError("empty query")
        return self._hash_to_vector((self.config.query_prefix or "") + query)

    def embed_chunks(self, chunks) -> None:
        for c, e in zip(chunks, self.embed_batch([c.content for c in chunks])):
            c.embedding = e

    def dimension(self) -> int:
        return self.dimension_

    def model_id(self) -> str:
        return self.model_id_
