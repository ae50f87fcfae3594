"""GPU: random walks over a store in the pipelined modes (trag_store_set_pipelined 1 and 2). Device-entry
searches -- single queries, whose selects overlap the next scan or ride in its launch, and batches, which take
other routes and wait for the pipeline -- are enqueued on two user streams with random k, between them small
inserts, removals, the bf16 copy coming and going, forced routes, host-entry searches, fences at random
moments and switches of the pipelining mode itself. Outputs are read only after a fence on the stream of the
call and a synchronisation; every one must equal the oracle's rows and score bits on the corpus of its own
moment. The cases of test_gpu_boundary_contract.py pin the known hand-overs; this looks for the ones nobody
listed."""
import os

import numpy as np
import pytest

from util import assert_same_ranking

pytestmark = pytest.mark.gpu

AUTO, EXACT, MIRROR = 0, 1, 5


@pytest.mark.parametrize("seed,n0,d,metric,pipe_mode", [
    (1, 40_000, 64, 0, 2), (2, 40_000, 64, 1, 1), (3, 300_000, 32, 2, 2), (4, 2_000_000, 32, 0, 2),
    (5, 2_000_000, 32, 1, 2), (6, 2_000_000, 32, 0, 1), (7, 70_000, 10, 0, 2), (8, 150_000, 128, 1, 2)])
def test_pipelined_walk(trag, oracle, seed, n0, d, metric, pipe_mode):
    import torch
    # TRAG_WALK_SEED shifts every walk to fresh random choices (for bug hunts; the default is what CI pins)
    rng = np.random.default_rng(7000 + seed + 100 * int(os.environ.get("TRAG_WALK_SEED", "0")))
    store = trag.VectorStore(trag.VectorStoreConfig(dimension=d, metric=trag.DistanceMetric(metric), device=0,
                                                    reserve_rows=n0 + 4096))
    store.fill_synthetic(n0, 900 + seed)
    base = store.read_rows(0, n0)
    added = []                       # small inserts, in order
    dead = np.zeros(n0, dtype=np.uint8)
    mirror = False
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    store.set_pipelined(pipe_mode)
    cur_mode = pipe_mode
    pending = []                     # (what, Q, k, n_rows then, dead then, output tensors or host results)
    steps = 70 if n0 <= 300_000 else 45

    def snapshot():
        return base.shape[0] + sum(a.shape[0] for a in added), dead.copy()

    for step in range(steps):
        op = rng.random()
        n = base.shape[0] + sum(a.shape[0] for a in added)
        if op < 0.08 and n < n0 + 3500:                  # a small insert (deferred to the next search's launch)
            m = int(rng.integers(1, 300))
            new = oracle.synth_rows(5000 + seed * 100 + step, 0, m, d)
            assert store.insert_rows(new) == n
            added.append(new)
            dead = np.concatenate([dead, np.zeros(m, dtype=np.uint8)])
        elif op < 0.16:                                  # removals
            for r in rng.integers(0, n, size=int(rng.integers(1, 4))):
                if not dead[r]:
                    store.remove_row(int(r))
                    dead[r] = 1
        elif op < 0.22 and d % 8 == 0 and n >= 65_536:   # the bf16 copy comes and goes
            mirror = not mirror
            store.set_bf16_mirror(mirror)
        elif op < 0.31:                                  # the pipelining mode itself
            cur_mode = int(rng.choice([0, 1, 2]))
            store.set_pipelined(cur_mode)
        elif op < 0.36:                                  # a fence in the middle
            store.fence(streams[int(rng.integers(2))].cuda_stream)
        elif op < 0.44:                                  # host entry (waits for the pipeline)
            k = int(rng.choice([1, 10, 50]))
            q = oracle.synth_rows(6000 + seed * 100 + step, 0, 1, d)
            store.set_scan_mode(AUTO)
            got = store.search_rows(q[0], k)
            pending.append((f"step {step} host entry k {k}", q, k) + snapshot() + (got,))
        else:                                            # device entry, one query or a batch
            nq = 1 if rng.random() < 0.8 else int(rng.choice([2, 3, 8, 20]))
            k = int(rng.choice([1, 5, 10, 50, 100]))
            Q = oracle.synth_rows(7000 + seed * 100 + step, 0, nq, d)
            if rng.random() < 0.2:
                r = int(rng.integers(base.shape[0]))
                Q[0] = base[r]                           # a stored row: a clear winner (or a removed one)
            # the route: whatever AUTO takes, the exact scan, or (single queries, with the copy) the scan of the copy
            store.set_scan_mode(int(rng.choice([AUTO, AUTO, EXACT] + ([MIRROR] if mirror and nq == 1 else []))))
            dq = torch.from_numpy(Q).cuda()
            out = (torch.full((nq, k), -1, dtype=torch.int32, device="cuda"), torch.zeros((nq, k), device="cuda"),
                   torch.zeros(nq, dtype=torch.int32, device="cuda"), torch.full((nq,), 9, dtype=torch.int32, device="cuda"))
            si = int(rng.integers(2))
            store.search_device(dq.data_ptr(), nq, k, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                                out[3].data_ptr(), streams[si].cuda_stream)
            pending.append((f"step {step} device entry nq {nq} k {k} stream {si} pipelined {cur_mode} mirror {mirror}", Q, k)
                           + snapshot() + ((dq, out, si),))
    for s in streams:
        store.fence(s.cuda_stream)
    torch.cuda.synchronize()
    full = np.concatenate([base] + added) if added else base
    for what, Q, k, n_then, dead_then, got in pending:
        corpus = full[:n_then]
        live = int(n_then - dead_then.sum())
        for b in range(Q.shape[0]):
            er, es = oracle.search(corpus, Q[b], k, metric, dead=dead_then)
            if isinstance(got[0], np.ndarray):
                rows, scores = got
            else:
                _, out, _ = got
                assert int(out[3][b]) == 0, (what, "status", int(out[3][b]))
                c = int(out[2][b])
                assert c == min(k, live), (what, c)
                rows, scores = out[0][b, :c].cpu().numpy().astype(np.uint32), out[1][b, :c].cpu().numpy()
            assert_same_ranking(rows, scores, er, es, f"walk {seed}: {what} q{b}")
    store.close()
