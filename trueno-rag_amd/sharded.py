"""Row-sharded search across the GPUs of one node: one process per GPU.

Rank g owns the contiguous global rows [g*ceil(N/G), min(N, (g+1)*ceil(N/G))).
Every rank scans its shard locally (no data-path collective); the per-shard
top-k lists, with GLOBAL row ids, are exchanged with ONE all-gather
(torch.distributed: RCCL on GPUs, gloo in the CPU tests) and merged on every
rank with the same ordering rule as a single-store search. No counterpart in
the reference, which is single-process.
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
    """All-gather + merge of per-shard top-k lists for up to `slots` queries in
    flight. Buffers are allocated once; `exchange()` allocates nothing."""

    def __init__(self, k: int, slots: int = 1, device=None, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.k, self.slots = k, slots
        self.device = device if device is not None else torch.device("cpu")
        z = lambda *shape, dt=torch.int32: torch.zeros(shape, dtype=dt, device=self.device)
        # wire format per rank and query: k score bit patterns, k global rows, 1 count
        self.pack = z(slots, 2 * k + 1)
        self.gathered = z(slots, self.world, 2 * k + 1)
        self.g_rows = z(slots, self.world, k)
        self.g_scores = z(slots, self.world, k, dt=torch.float32)
        self.g_n = z(slots, self.world)
        self.out_rows = z(slots, k)
        self.out_scores = z(slots, k, dt=torch.float32)
        self.out_n = z(slots)

    def exchange(self, slot, local_rows, local_scores, local_n, row0, stream_ptr=0):
        """local_*: this rank's result for the query (k entries, local row ids,
        local_n valid). Leaves the merged global top-k in out_rows/out_scores/out_n[slot]."""
        t, k = self.torch, self.k
        p = self.pack[slot]
        p[:k] = local_scores.view(t.int32)
        p[k:2 * k] = local_rows + row0
        p[2 * k:] = local_n
        if self.world > 1:
            self.dist.all_gather_into_tensor(self.gathered[slot].view(-1), p, group=self.group)
        else:
            self.gathered[slot, 0] = p
        g = self.gathered[slot]
        self.g_scores[slot] = g[:, :k].view(t.float32)
        self.g_rows[slot] = g[:, k:2 * k]
        self.g_n[slot] = g[:, 2 * k]
        L = _abi.lib()
        if self.device.type == "cuda":
            _abi.check(L.trag_merge_topk_device(
                C.c_void_p(self.g_rows[slot].data_ptr()), C.c_void_p(self.g_scores[slot].data_ptr()),
                C.c_void_p(self.g_n[slot].data_ptr()), self.world, k, k,
                C.c_void_p(self.out_rows[slot].data_ptr()), C.c_void_p(self.out_scores[slot].data_ptr()),
                C.c_void_p(self.out_n[slot].data_ptr()), C.c_void_p(stream_ptr) if stream_ptr else None))
        else:
            rows = np.ascontiguousarray(self.g_rows[slot].numpy().astype(np.uint32))
            scores = np.ascontiguousarray(self.g_scores[slot].numpy())
            cnt = np.ascontiguousarray(self.g_n[slot].numpy().astype(np.uint32))
            o_r, o_s, o_n = np.zeros(k, np.uint32), np.zeros(k, np.float32), C.c_uint32()
            u32p, fp = C.POINTER(C.c_uint32), C.POINTER(C.c_float)
            _abi.check(L.trag_merge_topk(rows.ctypes.data_as(u32p), scores.ctypes.data_as(fp),
                                         cnt.ctypes.data_as(u32p), self.world, k, k,
                                         o_r.ctypes.data_as(u32p), o_s.ctypes.data_as(fp), C.byref(o_n)))
            self.out_rows[slot] = t.from_numpy(o_r.astype(np.int32))
            self.out_scores[slot] = t.from_numpy(o_s)
            self.out_n[slot] = int(o_n.value)
        return self.out_rows[slot], self.out_scores[slot], self.out_n[slot]
