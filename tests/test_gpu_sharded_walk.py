"""GPU: random walks through a shard group (RCCL at world size 1, so pack, all-gather and merge really run). One
group serves TWO local stores in turn -- in overlap mode it switches each into the deferred-select mode and owes
at most one exchange at a time, which a call for the other store, another stream, a batch, a host-entry call or a
fence must settle -- with single-query and batch calls on two streams, k changing from call to call, removals
and small inserts between calls, a bf16 copy coming and going, fences at random moments. Nothing is synchronised
until the end; then every call's output must hold the oracle's global rows and score bits for the store as it
was when the call was made."""
import os

import numpy as np
import pytest

from conftest import has_gpu
from util import assert_same_ranking

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs a GPU")]


class Shadow:
    def __init__(self, trag, oracle, n, d, metric, seed, row0, row_mul):
        self.store = trag.VectorStore(trag.VectorStoreConfig(dimension=d, metric=trag.DistanceMetric(metric), device=0,
                                                             reserve_rows=n + 2048))
        self.store.fill_synthetic(n, seed)
        self.base = self.store.read_rows(0, n)
        self.added = []
        self.dead = np.zeros(n, dtype=np.uint8)
        self.row0, self.row_mul = row0, row_mul
        self.mirror = False

    def rows(self):
        return self.base.shape[0] + sum(a.shape[0] for a in self.added)

    def snapshot(self):
        return self.rows(), self.dead.copy()

    def corpus(self):
        return np.concatenate([self.base] + self.added) if self.added else self.base


@pytest.mark.parametrize("seed,metric,overlap", [(1, 0, True), (2, 1, True), (3, 2, True), (4, 0, False), (5, 1, True),
                                                 (6, 0, True)])
def test_sharded_walk(trag, oracle, seed, metric, overlap):
    import torch
    from trueno_rag_amd.sharded import ShardGroup
    rng = np.random.default_rng(9000 + seed + 100 * int(os.environ.get("TRAG_WALK_SEED", "0")))
    d, kmax = 64, 50
    shards = [Shadow(trag, oracle, 120_000, d, metric, 300 + seed, 5_000_000, 1),
              Shadow(trag, oracle, 70_000, d, metric, 400 + seed, 3, 7)]
    g = ShardGroup(0, 1, 0, k_max=kmax, max_queries=8, overlap=overlap)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    pending = []
    for step in range(90):
        op = rng.random()
        sh = shards[0] if rng.random() < 0.7 else shards[1]
        n = sh.rows()
        if op < 0.06 and n < sh.base.shape[0] + 1800:
            m = int(rng.integers(1, 200))
            new = oracle.synth_rows(8000 + seed * 100 + step, 0, m, d)
            assert sh.store.insert_rows(new) == n
            sh.added.append(new)
            sh.dead = np.concatenate([sh.dead, np.zeros(m, dtype=np.uint8)])
        elif op < 0.12:
            for r in rng.integers(0, n, size=int(rng.integers(1, 4))):
                if not sh.dead[r]:
                    sh.store.remove_row(int(r))
                    sh.dead[r] = 1
        elif op < 0.17:
            sh.mirror = not sh.mirror
            sh.store.set_bf16_mirror(sh.mirror)
        elif op < 0.22:
            g.fence(streams[int(rng.integers(2))].cuda_stream)
        elif op < 0.28:                                   # host entry through the group (synchronous)
            k = int(rng.choice([1, 10, kmax]))
            Q = oracle.synth_rows(8500 + seed * 100 + step, 0, int(rng.choice([1, 3, 11])), d)
            got = g.search_batch_rows(sh.store, sh.row0, sh.row_mul, Q, k)
            pending.append((f"step {step} host entry", sh, Q, k) + sh.snapshot() + (got,))
        else:
            nq = 1 if rng.random() < 0.8 else int(rng.choice([2, 5, 8]))
            k = int(rng.choice([1, 3, 10, 20, kmax]))
            Q = oracle.synth_rows(8700 + seed * 100 + step, 0, nq, d)
            dq = torch.from_numpy(Q).cuda()
            out = (torch.full((nq, k), -1, dtype=torch.int32, device="cuda"), torch.zeros((nq, k), device="cuda"),
                   torch.zeros(nq, dtype=torch.int32, device="cuda"), torch.full((nq,), 9, dtype=torch.int32, device="cuda"))
            si = int(rng.integers(2)) if rng.random() < 0.3 else 0
            g.search_device(sh.store, sh.row0, sh.row_mul, dq.data_ptr(), nq, k, out[0].data_ptr(), out[1].data_ptr(),
                            out[2].data_ptr(), out[3].data_ptr(), streams[si].cuda_stream)
            pending.append((f"step {step} device entry store {shards.index(sh)} nq {nq} k {k} stream {si} mirror {sh.mirror}",
                            sh, Q, k) + sh.snapshot() + ((dq, out),))
    for s in streams:
        g.fence(s.cuda_stream)
    torch.cuda.synchronize()
    full = {id(sh): sh.corpus() for sh in shards}
    for what, sh, Q, k, n_then, dead_then, got in pending:
        corpus = full[id(sh)][:n_then]
        for b in range(Q.shape[0]):
            er, es = oracle.search(corpus, Q[b], k, metric, dead=dead_then)
            want_rows = (er.astype(np.uint64) * sh.row_mul + sh.row0).astype(np.uint32)
            if isinstance(got[0], np.ndarray):
                rows, scores, c = got[0][b], got[1][b], int(got[2][b])
                rows, scores = rows[:c], scores[:c]
            else:
                out = got[1]
                assert int(out[3][b]) == 0, (what, "status", int(out[3][b]))
                c = int(out[2][b])
                rows, scores = out[0][b, :c].cpu().numpy().astype(np.uint32), out[1][b, :c].cpu().numpy()
            assert c == len(er), (what, c, len(er))
            assert_same_ranking(rows, scores, want_rows, es, f"walk {seed}: {what} q{b}")
    g.close()
    for sh in shards:
        sh.store.close()
