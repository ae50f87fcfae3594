"""GPU: scans long enough to skip the key stores of tiles no later stage can read (28 or more
tiles per scan wave: 2.7 M rows here) must return exactly what they return with every key
stored, and what the oracle returns. The corpora put the best rows where the bound is weakest:
all in the tiles of one scan wave, all in a wave's first or last tiles, in tie groups that
straddle tiles, next to removed rows."""
import os

import numpy as np
import pytest

from util import assert_same_ranking

pytestmark = pytest.mark.gpu

EXACT, MIRROR = 1, 5
N, D = 2_700_000, 128
WAVES = 1024          # scan waves of a full grid: tile t is scanned by wave t % 1024


def planted_corpus(rng, metric):
    corpus = rng.standard_normal((N, D)).astype(np.float32)
    if metric == 0:
        corpus /= np.linalg.norm(corpus, axis=1, keepdims=True).astype(np.float32)
    q = rng.standard_normal(D).astype(np.float32)
    q /= np.float32(np.linalg.norm(q))
    near = lambda a: (np.float32(a) * q + np.float32(1 - a) * rng.standard_normal(D).astype(np.float32) /
                      np.float32(np.sqrt(D))).astype(np.float32)
    n_tiles = N // 64
    # 30 strong rows in tiles of wave 7 only (7, 7 + 1024, ...): one published key for all of them
    for j in range(30):
        corpus[(7 + WAVES * j) * 64 + (j % 64)] = near(0.9 - 0.01 * j)
    # strong rows in the very first and the very last tile of other waves
    corpus[100 * 64 + 3] = near(0.95)
    corpus[(n_tiles - 1) * 64 + 63] = near(0.93)
    corpus[N - 1] = near(0.91)
    # a tie group of 12 identical rows spread over tiles of different waves
    tie = near(0.85)
    for j in range(12):
        corpus[(2000 + 37 * j) * 64 + 5] = tie
    return np.ascontiguousarray(corpus), q


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_sparse_key_stores_change_nothing(trag, oracle, metric):
    rng = np.random.default_rng(77 + metric)
    corpus, q0 = planted_corpus(rng, metric)
    dead = np.zeros(N, dtype=np.uint8)
    removed = [(7 + WAVES * 2) * 64 + 2, 100 * 64 + 3, (2000 + 37 * 3) * 64 + 5, 12345]
    queries = [q0, rng.standard_normal(D).astype(np.float32), corpus[(2000 + 37 * 5) * 64 + 5].copy(),
               np.zeros(D, dtype=np.float32)]
    ks = (1, 10, 40, 64, 65, 100)
    want = {}
    gathered = {}
    for sparse in ("1", "0"):
        os.environ["TRAG_SPARSE_KEYS"] = sparse      # read by a store at its first search
        try:
            store = trag.VectorStore(trag.VectorStoreConfig(dimension=D, metric=trag.DistanceMetric(metric),
                                                            reserve_rows=N))
            store.insert_rows(corpus)
            for r in removed:
                store.remove_row(r)
                dead[r] = 1
            store.set_bf16_mirror(True)
            for mode in (EXACT, MIRROR):
                store.set_scan_mode(mode)
                for qi, q in enumerate(queries):
                    for k in ks:
                        if mode == MIRROR and qi == 3 and k > 10:
                            continue                  # (a zero query goes back to the exact scan every time)
                        rows, scores = store.search_rows(q, k)
                        key = (qi, k)
                        if key not in want:
                            want[key] = oracle.search(corpus, q, k, metric, dead=dead)
                        er, es = want[key]
                        if qi == 1 and k == 10:
                            gathered[sparse, mode] = store.last_stats()["keys_kept_from"]
                        assert_same_ranking(rows, scores, er, es,
                                            f"sparse={sparse} mode={mode} metric={metric} query {qi} k={k}")
            # alternating queries whose best rows sit in different tiles: each leaves keys in keys[]
            # that would beat everything in the other's answer if a skipped tile were ever read
            store.set_scan_mode(0)
            picks = [5 * 64 + 1, 900_000, 2_222_222, N - 70]
            for rep in range(3):
                for r in picks:
                    q = corpus[r] + np.float32(1e-3) * rng.standard_normal(D).astype(np.float32)
                    rows, scores = store.search_rows(q, 10)
                    er, es = oracle.search(corpus, q, 10, metric, dead=dead)
                    assert_same_ranking(rows, scores, er, es, f"sparse={sparse} alternating row {r} metric={metric}")
            # TRAG_SPARSE_HUNT=n: n more random queries per scan (bug hunts; off by default)
            hunt = np.random.default_rng(4242 + metric)
            for i in range(int(os.environ.get("TRAG_SPARSE_HUNT", "0"))):
                q = hunt.standard_normal(D).astype(np.float32)
                if i % 5 == 0:
                    q = corpus[int(hunt.integers(N))] + np.float32(0.05) * q
                kk = int(hunt.choice([1, 5, 10, 33, 64]))
                er, es = oracle.search(corpus, q, kk, metric, dead=dead)
                for mode in (EXACT, MIRROR):
                    store.set_scan_mode(mode)
                    rows, scores = store.search_rows(q, kk)
                    assert_same_ranking(rows, scores, er, es, f"hunt {i} sparse={sparse} mode={mode} metric={metric} k={kk}")
            store.close()
        finally:
            os.environ.pop("TRAG_SPARSE_KEYS", None)
    # the switch did something: a bound was in force for a random query, in both scans, and none without
    assert gathered["1", EXACT] > 0x007FFFFF and gathered["1", MIRROR] > 0x007FFFFF, gathered
    assert gathered["0", EXACT] == 0 and gathered["0", MIRROR] == 0, gathered


@pytest.mark.parametrize("pipe_mode", [0, 1, 2])
def test_euclidean_long_store_finishes_queries_enqueued_far_ahead(trag, oracle, pipe_mode):
    """A 2.7 M-row Euclidean store of normalised vectors: nearly every distance shares the coarse score
    bin of the k-th best, which the one-workgroup select used to hand back (status 3) until the handle
    had learned the multi-level plan -- every query enqueued before that came back unfinished. With the
    sparse key stores the scan's own bound cuts the crowd: 200 queries enqueued ahead of the GPU, on a
    fresh handle, all finish, with the oracle's rows and score bits -- with the select behind its scan
    (0), on the handle's side stream (1), or riding in the next scan's launch (2: the bound it reads is its
    own scratch set's, tagged with its own scan's epoch, while the next scan publishes into the other)."""
    import torch
    n, d, k, B = 2_700_000, 64, 10, 200
    st = trag.VectorStore(trag.VectorStoreConfig(dimension=d, metric=trag.DistanceMetric(1), device=0, reserve_rows=n))
    st.fill_synthetic(n, 31)
    Q = oracle.synth_rows(32, 0, B, d)
    dq = torch.from_numpy(Q).cuda()
    rows = torch.zeros((B, k), dtype=torch.int32, device="cuda")
    scores = torch.zeros((B, k), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
    status = torch.full((B,), 9, dtype=torch.int32, device="cuda")
    sp = torch.cuda.current_stream().cuda_stream
    st.set_pipelined(pipe_mode)
    for b in range(B):      # one query per call, nothing synchronised in between
        st.search_device(dq.data_ptr() + b * d * 4, 1, k, rows.data_ptr() + b * k * 4, scores.data_ptr() + b * k * 4,
                         cnt.data_ptr() + b * 4, status.data_ptr() + b * 4, sp)
    st.fence(sp)
    torch.cuda.synchronize()
    assert int((status != 0).sum().item()) == 0, status.tolist()
    assert cnt.tolist() == [k] * B
    corpus = oracle.synth_rows(31, 0, n, d)
    for b in (0, 1, 99, 199):
        er, es = oracle.search(corpus, Q[b], k, 1)
        assert_same_ranking(rows[b].cpu().numpy().astype(np.uint32), scores[b].cpu().numpy(), er, es, f"query {b}")
    # the crowd really is there: without the bound the one-workgroup select gives up
    st.close()
