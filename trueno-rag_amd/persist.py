"""The reference CLI's on-disk index (crates/trueno-rag-cli/src/main.rs:133-155:
`<dir>/index.json` = PersistedIndex{chunks, embeddings, dimension,
embedder_type, model_name}) loaded into a device-resident VectorStore, and
the CLI's query loop (main.rs:479-492) served from it.

It is the only on-disk format of the reference that holds dense vectors
(pretty-printed JSON, one number per line)."""
import json
import os
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

from .index import DistanceMetric, VectorStore, VectorStoreConfig
from ._abi import DimensionMismatch


@dataclass
class PersistedChunk:
    """main.rs:148-154"""
    content: str
    title: Optional[str] = None
    source: Optional[str] = None


@dataclass
class PersistedIndex:
    """main.rs:133-146"""
    chunks: List[PersistedChunk] = field(default_factory=list)
    embeddings: np.ndarray = field(default_factory=lambda: np.zeros((0, 0), dtype=np.float32))
    dimension: int = 0
    embedder_type: str = ""          # #[serde(default)]
    model_name: Optional[str] = None  # #[serde(default)]

    @classmethod
    def load(cls, index_path: str) -> "PersistedIndex":
        index_file = os.path.join(index_path, "index.json")
        if not os.path.exists(index_file):
            raise FileNotFoundError(f"Index not found at: {index_file}")   # main.rs:420-422
        with open(index_file, "r", encoding="utf-8") as f:
            raw = json.load(f)
        dim = int(raw["dimension"])
        embs = raw["embeddings"]
        for e in embs:
            if len(e) != dim:
                raise DimensionMismatch(dim, len(e))
        arr = np.asarray(embs, dtype=np.float32).reshape(len(embs), dim)
        chunks = [PersistedChunk(c["content"], c.get("title"), c.get("source")) for c in raw["chunks"]]
        if len(chunks) != len(embs):
            raise ValueError(f"index.json holds {len(chunks)} chunks but {len(embs)} embeddings")
        return cls(chunks, arr, dim, raw.get("embedder_type", ""), raw.get("model_name"))

    def save(self, output: str) -> str:
        """main.rs:405-437 (serde_json::to_string_pretty: 2-space indent; every f32 is written
        in its shortest form that parses back to the same f32)."""
        os.makedirs(output, exist_ok=True)
        index_file = os.path.join(output, "index.json")

        def num(x):
            s = np.format_float_positional(np.float32(x), unique=True, trim="0")
            if abs(x) != 0 and (abs(x) >= 1e16 or abs(x) < 1e-5):
                s = np.format_float_scientific(np.float32(x), unique=True, trim="0")
            return s

        with open(index_file, "w", encoding="utf-8") as f:
            f.write("{\n  \"chunks\": [")
            for i, c in enumerate(self.chunks):
                f.write(("," if i else "") + "\n    {\n")
                f.write(f"      \"content\": {json.dumps(c.content, ensure_ascii=False)},\n")
                f.write(f"      \"title\": {json.dumps(c.title, ensure_ascii=False)},\n")
                f.write(f"      \"source\": {json.dumps(c.source, ensure_ascii=False)}\n    }}")
            f.write(("\n  " if self.chunks else "") + "],\n  \"embeddings\": [")
            for i, e in enumerate(np.asarray(self.embeddings, dtype=np.float32)):
                f.write(("," if i else "") + "\n    [\n")
                f.write(",\n".join("      " + num(float(x)) for x in e))
                f.write("\n    ]")
            f.write(("\n  " if len(self.embeddings) else "") + "],\n")
            f.write(f"  \"dimension\": {int(self.dimension)},\n")
            f.write(f"  \"embedder_type\": {json.dumps(self.embedder_type)},\n")
            f.write(f"  \"model_name\": {json.dumps(self.model_name)}\n}}")
        return index_file

    def to_store(self, device: int = -1) -> VectorStore:
        """Rows in index order, so that row i is chunk i (the CLI's `(i, sim)` pairs)."""
        store = VectorStore(VectorStoreConfig(dimension=self.dimension, metric=DistanceMetric.Cosine,
                                              device=device, reserve_rows=len(self.chunks)))
        if len(self.chunks):
            store.insert_rows(np.ascontiguousarray(self.embeddings, dtype=np.float32))
        return store


class IndexServer:
    """run_query's scoring loop (main.rs:479-492) on the GPU: cosine against every
    stored embedding, stable descending sort (equal scores: lower index first,
    which is exactly what the CLI's Vec-ordered loop gives), truncate to top_k."""

    def __init__(self, persisted: PersistedIndex, device: int = -1):
        self.persisted = persisted
        self.store = persisted.to_store(device)

    @classmethod
    def open(cls, index_path: str, device: int = -1) -> "IndexServer":
        return cls(PersistedIndex.load(index_path), device)

    def query(self, query_embedding, top_k: int) -> List[Tuple[int, float]]:
        q = np.ascontiguousarray(query_embedding, dtype=np.float32)
        if q.shape[0] != self.persisted.dimension:
            # the CLI's cosine_similarity returns 0.0 for every row on a length mismatch (main.rs:529-531)
            n = min(top_k, len(self.persisted.chunks))
            return [(i, 0.0) for i in range(n)]
        rows, scores = self.store.search_rows(q, top_k)
        return list(zip(rows.tolist(), scores.tolist()))

    def results_json(self, query_embedding, top_k: int):
        """The CLI's `--format json` records (main.rs:496-510)."""
        out = []
        for rank, (i, score) in enumerate(self.query(query_embedding, top_k)):
            c = self.persisted.chunks[i]
            out.append({"rank": rank + 1, "score": score, "content": c.content, "title": c.title,
                        "source": c.source})
        return out
