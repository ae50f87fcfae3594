"""Row-sharded search across the GPUs of one node: one process per GPU.

Rank g owns the contiguous global rows [g*ceil(N/G), min(N, (g+1)*ceil(N/G))).
Every rank scans its shard locally (no data-path collective); the per-shard
top-k lists, with GLOBAL row ids, are exchanged with ONE all-gather per batch
of queries (torch.distributed: RCCL on GPUs, gloo in the CPU tests) and merged
on every rank with the same ordering rule as a single-store search. No
counterpart in the reference, which is single-process.

Wire format per query: k score bit patterns, k global rows, 1 count (2k+1
32-bit words), written by trag_pack_topk_device and reduced by
trag_merge_packed_device; on CPU tensors the same is done with numpy and the
host merge trag_merge_topk.
"""
import ctypes as C

import numpy as np

from . import _abi


def shard_range(rank: int, world: int, n_total: int):
    """(first global row, row count) of a rank's shard."""
    per = (n_total + world - 1) // world
    row0 = min(rank * per, n_total)
    return row0, max(0, min(per, n_total - row0))


class ShardExchange:
    """All-gather + merge of per-shard top-k lists for batches of up to
    `max_queries` queries. Buffers are allocated once; exchange() allocates nothing."""

    def __init__(self, k: int, max_queries: int = 1, device=None, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.k, self.max_queries = k, max_queries
        self.rec = 2 * k + 1
        self.device = device if device is not None else torch.device("cpu")
        self.pack = torch.zeros(max_queries * self.rec, dtype=torch.int32, device=self.device)
        self.gathered = torch.zeros(self.world * max_queries * self.rec, dtype=torch.int32, device=self.device)

    def exchange(self, rows, scores, counts, row0, out_rows, out_scores, out_counts, stream_ptr=0):
        """rows/scores: [B, k] (int32 local row ids / float32), counts: [B] int32 —
        this rank's results. Writes the merged global top-k of every query into
        out_rows/out_scores [B, k] and out_counts [B]."""
        t, k, rec = self.torch, self.k, self.rec
        B = rows.shape[0] if rows.dim() == 2 else 1
        assert B <= self.max_queries
        pack = self.pack[:B * rec]
        gathered = self.gathered[:self.world * B * rec]
        L = _abi.lib()
        sp = C.c_void_p(stream_ptr) if stream_ptr else None
        if self.device.type == "cuda":
            _abi.check(L.trag_pack_topk_device(C.c_void_p(rows.data_ptr()), C.c_void_p(scores.data_ptr()),
                                               C.c_void_p(counts.data_ptr()), row0, B, k,
                                               C.c_void_p(pack.data_ptr()), sp))
        else:
            p = pack.view(B, rec)
            p[:, :k] = scores.reshape(B, k).view(t.int32)
            p[:, k:2 * k] = rows.reshape(B, k) + row0
            p[:, 2 * k] = counts.reshape(B)
        if self.world > 1:
            self.dist.all_gather_into_tensor(gathered, pack, group=self.group)
        else:
            gathered.copy_(pack)
        if self.device.type == "cuda":
            _abi.check(L.trag_merge_packed_device(C.c_void_p(gathered.data_ptr()), self.world, B, k,
                                                  C.c_void_p(out_rows.data_ptr()),
                                                  C.c_void_p(out_scores.data_ptr()),
                                                  C.c_void_p(out_counts.data_ptr()), sp))
        else:
            g = gathered.view(self.world, B, rec).numpy()
            u32p, fp = C.POINTER(C.c_uint32), C.POINTER(C.c_float)
            for b in range(B):
                g_scores = np.ascontiguousarray(g[:, b, :k]).view(np.float32)
                g_rows = np.ascontiguousarray(g[:, b, k:2 * k]).astype(np.uint32)
                g_n = np.ascontiguousarray(g[:, b, 2 * k]).astype(np.uint32)
                o_r, o_s, o_n = np.zeros(k, np.uint32), np.zeros(k, np.float32), C.c_uint32()
                _abi.check(L.trag_merge_topk(g_rows.ctypes.data_as(u32p), g_scores.ctypes.data_as(fp),
                                             g_n.ctypes.data_as(u32p), self.world, k, k,
                                             o_r.ctypes.data_as(u32p), o_s.ctypes.data_as(fp), C.byref(o_n)))
                out_rows.reshape(B, k)[b] = t.from_numpy(o_r.astype(np.int32))
                out_scores.reshape(B, k)[b] = t.from_numpy(o_s)
                out_counts.reshape(B)[b] = int(o_n.value)
