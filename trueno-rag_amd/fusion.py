"""FusionStrategy (src/fusion.rs:9-63 of the reference) over trag_fuse."""
import ctypes as C
from dataclasses import dataclass
from typing import Dict, Hashable, List, Sequence, Tuple

import numpy as np

from . import _abi

_RRF, _LINEAR, _CONVEX, _DBSF, _UNION, _INTERSECTION = range(6)


@dataclass
class FusionStrategy:
    """Externally tagged like the serde enum: kind in {RRF, Linear, Convex, DBSF,
    Union, Intersection}; `param` is k / dense_weight / alpha."""
    kind: str = "RRF"
    param: float = 60.0

    @classmethod
    def default(cls) -> "FusionStrategy":
        return cls("RRF", 60.0)  # src/fusion.rs:33-37

    @classmethod
    def RRF(cls, k: float = 60.0):
        return cls("RRF", k)

    @classmethod
    def Linear(cls, dense_weight: float):
        return cls("Linear", dense_weight)

    @classmethod
    def Convex(cls, alpha: float):
        return cls("Convex", alpha)

    @classmethod
    def DBSF(cls):
        return cls("DBSF", 0.0)

    @classmethod
    def Union(cls):
        return cls("Union", 0.0)

    @classmethod
    def Intersection(cls):
        return cls("Intersection", 0.0)

    def to_json(self):
        if self.kind == "RRF":
            return {"RRF": {"k": self.param}}
        if self.kind == "Linear":
            return {"Linear": {"dense_weight": self.param}}
        if self.kind == "Convex":
            return {"Convex": {"alpha": self.param}}
        return self.kind

    @classmethod
    def from_json(cls, obj):
        if isinstance(obj, str):
            return cls(obj, 0.0)
        (kind, body), = obj.items()
        return cls(kind, float(next(iter(body.values()))))

    def fuse(self, dense_results: Sequence[Tuple[Hashable, float]],
             sparse_results: Sequence[Tuple[Hashable, float]]) -> List[Tuple[Hashable, float]]:
        """src/fusion.rs:42-63. Ids may be any hashable (ChunkId, int, ...)."""
        code = {"RRF": _RRF, "Linear": _LINEAR, "Convex": _CONVEX, "DBSF": _DBSF, "Union": _UNION,
                "Intersection": _INTERSECTION}[self.kind]
        token: Dict[Hashable, int] = {}
        back: List[Hashable] = []

        def tok(i):
            t = token.get(i)
            if t is None:
                t = len(back)
                token[i] = t
                back.append(i)
            return t

        di = np.array([tok(i) for i, _ in dense_results], dtype=np.uint64)
        ds = np.array([s for _, s in dense_results], dtype=np.float32)
        si = np.array([tok(i) for i, _ in sparse_results], dtype=np.uint64)
        ss = np.array([s for _, s in sparse_results], dtype=np.float32)
        cap = max(di.size + si.size, 1)
        out_ids = np.empty(cap, dtype=np.uint64)
        out_scores = np.empty(cap, dtype=np.float32)
        n = C.c_uint32()
        u64p, fp = C.POINTER(C.c_uint64), C.POINTER(C.c_float)
        _abi.check(_abi.lib().trag_fuse(code, self.param, di.ctypes.data_as(u64p), ds.ctypes.data_as(fp), di.size,
                                        si.ctypes.data_as(u64p), ss.ctypes.data_as(fp), si.size,
                                        out_ids.ctypes.data_as(u64p), out_scores.ctypes.data_as(fp), C.byref(n)))
        return [(back[int(out_ids[i])], float(out_scores[i])) for i in range(n.value)]


def rrf_device(dense_ids_ptr: int, dense_n_ptr: int, dense_stride: int, sparse_ids_ptr: int, sparse_n_ptr: int,
               sparse_stride: int, n_queries: int, k: float, k_out: int, out_ids_ptr: int, out_scores_ptr: int,
               out_n_ptr: int, stream: int = 0) -> None:
    """RRF (src/fusion.rs:68-84) for a batch of queries whose dense and sparse lists are in device
    memory, as u32 ids of one id space; every argument named *_ptr is a device address. Enqueued on
    `stream`, no synchronisation (trag_fuse_rrf_device)."""
    _abi.check(_abi.lib().trag_fuse_rrf_device(dense_ids_ptr, dense_n_ptr or None, dense_stride, sparse_ids_ptr,
                                               sparse_n_ptr or None, sparse_stride, n_queries, k, k_out, out_ids_ptr,
                                               out_scores_ptr, out_n_ptr, stream or None))

