"""CPU, world size 2 (gloo): the sharded path's exchange step — shard ranges,
all-gather of per-shard top-k with global rows, merge — against a single
global oracle search. The per-shard lists come from the oracle (no GPU here);
on a GPU box the same ShardExchange runs over RCCL (bench.py --gpus N)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, d, k, out_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import load_package
    from oracle import oracle
    pkg = load_package()
    from trueno_rag_amd.sharded import ShardExchange, shard_range
    row0, n_local = shard_range(rank, world, n_total)
    corpus = oracle.synth_rows(4242, row0, n_local, d)           # this rank's slice of the global corpus
    corpus_dup = corpus.copy()
    queries = oracle.synth_rows(4343, 0, 3, d)
    ex = ShardExchange(k, max_queries=3)
    lr = torch.zeros((3, k), dtype=torch.int32)
    ls = torch.zeros((3, k), dtype=torch.float32)
    ln = torch.zeros(3, dtype=torch.int32)
    for qi in range(3):
        r, s = oracle.search(corpus_dup, queries[qi], k, 0)
        lr[qi, :len(r)] = torch.from_numpy(r.astype(np.int32))
        ls[qi, :len(s)] = torch.from_numpy(s)
        ln[qi] = len(r)
    o_r = torch.zeros((3, k), dtype=torch.int32)
    o_s = torch.zeros((3, k), dtype=torch.float32)
    o_n = torch.zeros(3, dtype=torch.int32)
    ex.exchange(lr, ls, ln, row0, o_r, o_s, o_n)          # one all-gather for the whole batch
    results = [(o_r[qi].numpy().copy(), o_s[qi].numpy().copy(), int(o_n[qi])) for qi in range(3)]
    out_q.put((rank, results))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total,k", [(5000, 10), (7, 10), (1001, 50)])
def test_two_rank_exchange_matches_global_search(oracle, n_total, k):
    world, d = 2, 64
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, d, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    corpus = oracle.synth_rows(4242, 0, n_total, d)
    queries = oracle.synth_rows(4343, 0, 3, d)
    for qi in range(3):
        er, es = oracle.search(corpus, queries[qi], k, 0)
        for rank in range(world):
            r, s, n = got[rank][qi]
            assert n == len(er)
            assert r[:n].astype(np.uint32).tolist() == er.tolist(), f"rank {rank} query {qi}"
            assert s[:n].tobytes() == es.tobytes()


def test_shard_ranges_cover_everything(trag):
    from trueno_rag_amd.sharded import shard_range
    for n in (0, 1, 7, 8, 10_000_000, 1_000_001):
        for world in (1, 2, 4, 8):
            spans = [shard_range(r, world, n) for r in range(world)]
            assert sum(c for _, c in spans) == n
            pos = 0
            for r0, c in spans:
                assert r0 == min(pos, n) or c == 0
                pos += c


class _OracleStore:
    """Stand-in for a rank's VectorStore on machines without a GPU: same row-level methods,
    answers from the CPU oracle."""

    def __init__(self, oracle, d):
        self.o, self.d = oracle, d
        self.rows = np.zeros((0, d), dtype=np.float32)
        self.dead = np.zeros(0, dtype=np.uint8)

    def insert_rows(self, rows):
        first = self.rows.shape[0]
        self.rows = np.concatenate([self.rows, rows])
        self.dead = np.concatenate([self.dead, np.zeros(rows.shape[0], np.uint8)])
        return first

    def remove_row(self, r):
        self.dead[r] = 1

    def len(self):
        return int((self.dead == 0).sum())

    def search_batch_rows(self, Q, k):
        B = Q.shape[0]
        rows = np.zeros((B, k), np.uint32)
        scores = np.zeros((B, k), np.float32)
        n = np.zeros(B, np.uint32)
        for b in range(B):
            if self.rows.shape[0]:
                r, s = self.o.search(self.rows, Q[b], k, 0, dead=self.dead)
                rows[b, :len(r)], scores[b, :len(s)], n[b] = r, s, len(r)
        return rows, scores, n


def _striped_worker(rank, world, port, out_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import load_package
    from oracle import oracle
    load_package()
    from trueno_rag_amd.sharded import ShardedVectorStore
    d = 32
    store = ShardedVectorStore(_OracleStore(oracle, d), max_queries=2)
    corpus = oracle.synth_rows(99, 0, 1003, d)
    corpus[500] = corpus[3]                                  # a tie across shards: lower global row first
    assert store.insert_rows(corpus[:1]) == 0                # uneven blocks: stripes must stay in step
    assert store.insert_rows(corpus[1:700]) == 1
    assert store.insert_rows(corpus[700:]) == 700
    for r in (3, 10, 11, 998):
        store.remove_row(r)
    n_live = store.len()
    Q = np.concatenate([corpus[3:4], oracle.synth_rows(100, 0, 4, d)])
    rows, scores, counts = store.search_batch_rows(Q, 20)
    out_q.put((rank, n_live, rows, scores, counts))
    dist.barrier()
    dist.destroy_process_group()


def test_striped_store_matches_a_single_store(oracle):
    world, d = 2, 32
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_striped_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    corpus = oracle.synth_rows(99, 0, 1003, d)
    corpus[500] = corpus[3]
    dead = np.zeros(1003, np.uint8)
    dead[[3, 10, 11, 998]] = 1
    Q = np.concatenate([corpus[3:4], oracle.synth_rows(100, 0, 4, d)])
    for rank, n_live, rows, scores, counts in got:
        assert n_live == 999
        for b in range(5):
            er, es = oracle.search(corpus, Q[b], 20, 0, dead=dead)
            assert counts[b] == 20
            assert rows[b].tolist() == er.tolist(), f"rank {rank} query {b}"
            assert scores[b].tobytes() == es.tobytes()
    assert got[0][2][0][0] == 500                            # row 3 removed: its duplicate, on the other shard, wins
