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
        # device records over a gloo group (several ranks sharing one GPU, tests): the gather is
        # staged through pinned host buffers, since gloo moves host memory
        self.host_staged = (self.device.type == "cuda" and self.world > 1
                            and dist.get_backend(group) == "gloo")
        if self.host_staged:
            self.h_pack = torch.zeros(max_queries * self.rec, dtype=torch.int32).pin_memory()
            self.h_gathered = torch.zeros(self.world * max_queries * self.rec, dtype=torch.int32).pin_memory()

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
        if self.host_staged:
            stream = t.cuda.ExternalStream(stream_ptr) if stream_ptr else t.cuda.current_stream()
            with t.cuda.stream(stream):
                hp, hg = self.h_pack[:B * rec], self.h_gathered[:self.world * B * rec]
                hp.copy_(pack, non_blocking=True)
                stream.synchronize()
                self.dist.all_gather_into_tensor(hg, hp, group=self.group)
                gathered.copy_(hg, non_blocking=True)
        elif self.world > 1:
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


class ShardedVectorStore:
    """A VectorStore striped over the ranks of a process group (one process per GPU).

    Global row g lives on rank g % world as local row g // world: appends keep every shard
    within one row of the others, global row ids never move, and ascending global row is
    insertion order, so the tie rule of a single store (equal scores: lower row first) carries
    over unchanged. Every rank calls every method with the same arguments (collective, like
    torch.distributed itself); search returns the same merged result on every rank.

    `local` is this rank's store: a trueno_rag_amd.VectorStore on its GPU, or any object with
    insert_rows / search_batch_rows / remove_row / len (the CPU tests use an oracle-backed one).
    No counterpart in the reference, which is single-process.
    """

    def __init__(self, local, k_max: int = 100, max_queries: int = 256, device=None, group=None):
        import torch.distributed as dist
        self.local = local
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_rows = 0                      # global rows ever inserted
        self.k_max, self.max_queries, self.device = k_max, max_queries, device
        self._ex = {}

    def _exchange(self, k):
        if k not in self._ex:
            self._ex[k] = ShardExchange(k, max_queries=self.max_queries, device=self.device, group=self.group)
        return self._ex[k]

    def insert_rows(self, rows) -> int:
        """Append an (n, dim) block (the same block on every rank); returns its first global row."""
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        g0 = self.n_rows
        first = (self.rank - g0) % self.world          # index of this rank's first row in the block
        mine = rows[first::self.world]
        if mine.shape[0]:
            local0 = self.local.insert_rows(mine)
            assert local0 == (g0 + first) // self.world, "local store out of step with the striping"
        self.n_rows += rows.shape[0]
        return g0

    def remove_row(self, global_row: int) -> None:
        if global_row % self.world == self.rank:
            self.local.remove_row(global_row // self.world)

    def len(self) -> int:
        import torch
        import torch.distributed as dist
        n = torch.tensor([int(self.local.len())], dtype=torch.int64,
                         device=self.device if self.device is not None else "cpu")
        if self.world > 1:
            dist.all_reduce(n, group=self.group)
        return int(n.item())

    def search_batch_rows(self, queries, k: int):
        """(global rows [B, k], scores [B, k], counts [B]) of the merged top k, on every rank."""
        import torch
        Q = np.ascontiguousarray(queries, dtype=np.float32)
        B = Q.shape[0]
        if k == 0 or B == 0:
            return (np.zeros((B, 0), np.uint32), np.zeros((B, 0), np.float32), np.zeros(B, np.uint32))
        rows, scores, counts = self.local.search_batch_rows(Q, k)
        dev = self.device if self.device is not None else torch.device("cpu")
        g = rows.astype(np.int64) * self.world + self.rank            # local row -> global row
        lr = torch.from_numpy(g.astype(np.int32)).to(dev)
        ls = torch.from_numpy(np.ascontiguousarray(scores, dtype=np.float32)).to(dev)
        ln = torch.from_numpy(counts.astype(np.int32)).to(dev)
        o_r = torch.zeros((B, k), dtype=torch.int32, device=dev)
        o_s = torch.zeros((B, k), dtype=torch.float32, device=dev)
        o_n = torch.zeros(B, dtype=torch.int32, device=dev)
        ex = self._exchange(k)
        for b0 in range(0, B, self.max_queries):
            b1 = min(B, b0 + self.max_queries)
            ex.exchange(lr[b0:b1], ls[b0:b1], ln[b0:b1], 0, o_r[b0:b1], o_s[b0:b1], o_n[b0:b1])
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        return (o_r.cpu().numpy().astype(np.uint32), o_s.cpu().numpy(), o_n.cpu().numpy().astype(np.uint32))

    def search_rows(self, query, k: int):
        r, s, n = self.search_batch_rows(np.asarray(query, dtype=np.float32)[None, :], k)
        return r[0, :n[0]], s[0, :n[0]]
