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
        elif self.dist.is_initialized():
            # (also with one rank: the collective still runs, RCCL on a GPU)
            if stream_ptr and self.device.type == "cuda":
                with t.cuda.stream(t.cuda.ExternalStream(stream_ptr)):
                    self.dist.all_gather_into_tensor(gathered, pack, group=self.group)
            else:
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
                g_n = (np.ascontiguousarray(g[:, b, 2 * k]) & 0xFFFF).astype(np.uint32)  # (bits 16..23: status)
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


class ShardGroup:
    """One rank of a row-sharded search through the C ABI (trag_shard_group_*): the library
    itself runs pack -> RCCL all-gather -> merge, one C call per step. The ranks meet through an
    RCCL unique id that rank 0 creates and `share_id` carries to the others (by default over the
    torch.distributed default group, which may be gloo: the id is 128 bytes of host memory)."""

    def __init__(self, rank: int, world: int, device: int, k_max: int, max_queries: int = 1,
                 overlap: bool = False, unique_id: bytes = None, share_id=None):
        L = _abi.lib()
        if unique_id is None:
            buf = (C.c_uint8 * _abi.SHARD_ID_BYTES)()
            if rank == 0:
                _abi.check(L.trag_shard_unique_id(buf))
            raw = bytes(buf)
            if world > 1:
                raw = (share_id or self._broadcast_id)(raw, rank)
            unique_id = raw
        assert len(unique_id) == _abi.SHARD_ID_BYTES
        self.rank, self.world, self.k_max, self.max_queries = rank, world, k_max, max_queries
        cfg = _abi.ShardConfig(rank, world, device, k_max, max_queries, 1 if overlap else 0)
        idbuf = (C.c_uint8 * _abi.SHARD_ID_BYTES).from_buffer_copy(unique_id)
        self._h = _abi._H()
        _abi.check(L.trag_shard_group_create(C.byref(cfg), idbuf, C.byref(self._h)))

    @staticmethod
    def _broadcast_id(raw: bytes, rank: int) -> bytes:
        import torch.distributed as dist
        box = [raw if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _abi.lib().trag_shard_group_destroy(self._h)
            self._h = _abi._H()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def search_device(self, store, row0: int, row_mul: int, d_queries_ptr: int, n_queries: int, k: int,
                      d_rows_ptr: int, d_scores_ptr: int, d_n_ptr: int, d_status_ptr: int = 0, stream: int = 0):
        """Collective. Local search on `store` + exchange + merge, enqueued on `stream`; every rank's
        output arrays receive the global top k (global row = row0 + local row * row_mul)."""
        dim = store.config().dimension
        _abi.check(_abi.lib().trag_shard_search_device(
            self._h, store.handle, row0, row_mul, C.c_void_p(d_queries_ptr), n_queries, dim, k,
            C.c_void_p(d_rows_ptr), C.c_void_p(d_scores_ptr), C.c_void_p(d_n_ptr),
            C.c_void_p(d_status_ptr) if d_status_ptr else None, C.c_void_p(stream) if stream else None), dim, dim)

    def info(self):
        """What the RCCL communicator says about itself: ranks it spans, this rank, its device."""
        w, r, d = C.c_uint32(), C.c_uint32(), C.c_int32()
        _abi.check(_abi.lib().trag_shard_group_info(self._h, C.byref(w), C.byref(r), C.byref(d)))
        return {"comm_world": w.value, "comm_rank": r.value, "comm_device": d.value}

    def fence(self, stream: int = 0):
        _abi.check(_abi.lib().trag_shard_fence(self._h, C.c_void_p(stream) if stream else None))

    def search_batch_rows(self, store, row0: int, row_mul: int, queries, k: int):
        """Collective, host arrays, synchronous: (global rows [B, k], scores [B, k], counts [B])."""
        Q = np.ascontiguousarray(queries, dtype=np.float32)
        B, dim = Q.shape
        rows = np.zeros((B, max(k, 1)), dtype=np.uint32)
        scores = np.zeros((B, max(k, 1)), dtype=np.float32)
        n = np.zeros(B, dtype=np.uint32)
        u32p, fp = C.POINTER(C.c_uint32), C.POINTER(C.c_float)
        _abi.check(_abi.lib().trag_shard_search_batch(
            self._h, store.handle, row0, row_mul, Q.ctypes.data_as(fp), B, dim, k, rows.ctypes.data_as(u32p),
            scores.ctypes.data_as(fp), n.ctypes.data_as(u32p)), store.config().dimension, dim)
        return rows[:, :k], scores[:, :k], n


def rccl_version() -> int:
    v = C.c_int(0)
    _abi.check(_abi.lib().trag_shard_rccl_version(C.byref(v)))
    return v.value


class MultiGpuVectorStore:
    """One process, several GPUs (trag_sharded_*): a store striped by rows over `devices`, one RCCL
    communicator over all of them (exchange="rccl"), or plain device-to-device copies
    (exchange="copy": no RCCL needed, and the same GPU may be named twice). Row-level surface of
    VectorStore: insert_rows / search_rows / search_batch_rows / remove_row / len."""

    def __init__(self, config, devices, exchange: str = "rccl"):
        self._config = config
        cfg = _abi.Config(config.dimension, int(config.metric), -1, config.scan_mode, config.reserve_rows)
        ids = (C.c_int32 * len(devices))(*devices)
        self._h = _abi._H()
        _abi.check(_abi.lib().trag_sharded_create(C.byref(cfg), len(devices), ids,
                                                  {"rccl": 0, "copy": 1}[exchange], C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _abi.lib().trag_sharded_destroy(self._h)
            self._h = _abi._H()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def config(self):
        return self._config

    def insert_rows(self, rows) -> int:
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        first = C.c_uint32()
        _abi.check(_abi.lib().trag_sharded_insert(self._h, rows.ctypes.data_as(C.POINTER(C.c_float)), rows.shape[0],
                                                  rows.shape[1], C.byref(first)), self._config.dimension, rows.shape[1])
        return first.value

    def fill_synthetic(self, n: int, seed: int) -> int:
        first = C.c_uint32()
        _abi.check(_abi.lib().trag_sharded_fill_synthetic(self._h, n, seed, C.byref(first)))
        return first.value

    def overwrite_row(self, row: int, vec) -> None:
        v = np.ascontiguousarray(vec, dtype=np.float32).reshape(-1)
        _abi.check(_abi.lib().trag_sharded_overwrite(self._h, row, v.ctypes.data_as(C.POINTER(C.c_float)), v.size),
                   self._config.dimension, v.size)

    def remove_row(self, row: int) -> None:
        _abi.check(_abi.lib().trag_sharded_remove(self._h, row))

    def len(self) -> int:
        return int(_abi.lib().trag_sharded_len(self._h))

    def rows(self) -> int:
        return int(_abi.lib().trag_sharded_rows(self._h))

    def shard_handle(self, i: int):
        return _abi._H(_abi.lib().trag_sharded_shard(self._h, i))

    def set_scan_mode(self, mode: int) -> None:
        for i in range(_abi.lib().trag_sharded_devices(self._h)):
            _abi.check(_abi.lib().trag_store_set_scan_mode(self.shard_handle(i), mode))

    def search_batch_rows(self, queries, k: int):
        Q = np.ascontiguousarray(queries, dtype=np.float32)
        if Q.ndim != 2:
            raise _abi.InvalidConfig("queries must be a 2-D array")
        B, dim = Q.shape
        k = min(int(k), self.len())
        rows = np.zeros((B, max(k, 1)), dtype=np.uint32)
        scores = np.zeros((B, max(k, 1)), dtype=np.float32)
        n = np.zeros(B, dtype=np.uint32)
        u32p, fp = C.POINTER(C.c_uint32), C.POINTER(C.c_float)
        _abi.check(_abi.lib().trag_sharded_search_batch(self._h, Q.ctypes.data_as(fp), B, dim, k,
                                                        rows.ctypes.data_as(u32p), scores.ctypes.data_as(fp),
                                                        n.ctypes.data_as(u32p)), self._config.dimension, dim)
        return rows[:, :k], scores[:, :k], n

    def search_rows(self, query, k: int):
        r, s, n = self.search_batch_rows(np.asarray(query, dtype=np.float32)[None, :], k)
        return r[0, :n[0]].copy(), s[0, :n[0]].copy()
