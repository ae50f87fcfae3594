"""Rerankers of the reference that take the retrieval path's results
(src/rerank.rs:6-124, 250-300): host-side string features, no GPU work."""
import copy
from dataclasses import dataclass
from typing import List

import numpy as np

from .retrieve import RetrievalResult

_f = np.float32


@dataclass
class LexicalReranker:
    """src/rerank.rs:17-122. Arithmetic in f32, in the reference's order."""
    exact_match_weight: float = 0.3
    coverage_weight: float = 0.5
    position_weight: float = 0.2
    case_insensitive: bool = True

    @classmethod
    def new(cls) -> "LexicalReranker":
        return cls()

    def with_weights(self, exact_match: float, coverage: float, position: float) -> "LexicalReranker":
        self.exact_match_weight, self.coverage_weight, self.position_weight = exact_match, coverage, position
        return self

    def score(self, query: str, content: str) -> float:
        if self.case_insensitive:
            query, content = query.lower(), content.lower()
        terms = query.split()
        if not terms:
            return 0.0
        exact = _f(1.0) if query in content else _f(0.0)
        n = _f(len(terms))
        coverage = _f(sum(1 for t in terms if t in content)) / n
        cbytes = content.encode("utf-8")
        pos_sum = _f(0.0)
        for t in terms:
            pos = cbytes.find(t.encode("utf-8"))      # str::find returns a BYTE offset
            if pos >= 0:
                pos_sum = _f(pos_sum + _f(1.0) / (_f(1.0) + _f(pos) / _f(100.0)))
        position = pos_sum / n
        return float(_f(_f(_f(self.exact_match_weight) * exact + _f(self.coverage_weight) * coverage)
                        + _f(self.position_weight) * position))

    def rerank(self, query: str, candidates: List[RetrievalResult], top_k: int) -> List[RetrievalResult]:
        scored = [(c, self.score(query, c.chunk.content)) for c in candidates]
        scored.sort(key=lambda p: -p[1])               # stable, descending
        out = []
        for c, s in scored[:top_k]:
            r = copy.copy(c)
            r.rerank_score = s
            out.append(r)
        return out


class NoOpReranker:
    """src/rerank.rs: passes candidates through, cut to top_k."""

    def rerank(self, query: str, candidates: List[RetrievalResult], top_k: int) -> List[RetrievalResult]:
        return list(candidates[:top_k])
