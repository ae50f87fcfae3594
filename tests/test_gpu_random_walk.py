"""GPU: random walks over the store's whole surface. Each walk interleaves inserts (with
awkward rows), overwrites, removals, growth past the thresholds at which the matrix-core and
mirror routes switch on, mirror on/off, and searches in every scan mode, single and batched,
with random k; a host shadow of the corpus and the tombstones feeds the oracle, and every
result must equal the oracle's rows and score bits. The cases of the other test files pin
known corners; this one looks for the interactions nobody listed."""
import ctypes as C
import os

import numpy as np
import pytest

from util import assert_same_ranking

pytestmark = pytest.mark.gpu

AUTO, EXACT, STREAM, MFMA, BF16, MIRROR = 0, 1, 2, 3, 4, 5


def awkward_rows(rng, m, d, shadow_rows):
    """m new rows: mostly Gaussian, some copies of rows already stored (ties), zeros, tiny, huge."""
    rows = rng.standard_normal((m, d)).astype(np.float32)
    for i in range(m):
        u = rng.random()
        if u < 0.03 and shadow_rows:
            rows[i] = shadow_rows[int(rng.integers(len(shadow_rows)))]      # exact duplicate of a stored row
        elif u < 0.04:
            rows[i] = 0.0
        elif u < 0.05:
            rows[i] *= np.float32(1e-20)
        elif u < 0.06:
            rows[i] *= np.float32(1e6)
        elif u < 0.10:
            rows[i] = np.round(rows[i] * 2) / 2                             # coarse values: many equal scores
    return np.ascontiguousarray(rows)


def modes_for(n_live, d, k, batch, mirror):
    """scan modes this call may be forced into (AUTO is always allowed)."""
    modes = [AUTO, EXACT]
    if k <= 4000:
        modes.append(STREAM)
    if batch and d % 32 == 0 and k <= 4000:
        modes.append(MFMA)
    if batch and d % 64 == 0 and n_live >= 65_536 and k <= 256:
        modes.append(BF16)
    if mirror and not batch and n_live >= 65_536 and k <= 256:
        modes.append(MIRROR)
    return modes


@pytest.mark.parametrize("seed,d,metric,n0", [(1, 128, 0, 60_000), (2, 384, 1, 64_000), (3, 64, 2, 66_000),
                                               (4, 256, 0, 70_000), (5, 100, 1, 3_000), (6, 384, 2, 65_000),
                                               (7, 768, 0, 65_000), (8, 7, 0, 500), (9, 384, 0, 66_000),
                                               (10, 128, 1, 65_500), (11, 512, 2, 65_000), (12, 64, 0, 131_000),
                                               (13, 1024, 0, 65_600), (14, 36, 1, 20_000), (15, 192, 2, 70_000),
                                               (16, 4096, 0, 65_536)])
def test_random_walk(trag, oracle, seed, d, metric, n0):
    # TRAG_WALK_SEED shifts every walk to fresh random choices (for bug hunts; the default is what CI pins)
    rng = np.random.default_rng(1000 + seed + 100 * int(os.environ.get("TRAG_WALK_SEED", "0")))
    store = trag.VectorStore(trag.VectorStoreConfig(dimension=d, metric=trag.DistanceMetric(metric)))
    L = trag.lib()
    fp = C.POINTER(C.c_float)
    first = awkward_rows(rng, n0, d, [])
    store.insert_rows(first)
    corpus = first
    dead = np.zeros(n0, dtype=np.uint8)
    mirror = False
    searches = 0
    for step in range(120):
        op = rng.random()
        n = corpus.shape[0]
        if op < 0.12:                                   # grow (crosses 65 536 rows in most walks)
            m = int(rng.integers(1, 4000))
            new = awkward_rows(rng, m, d, [corpus[int(rng.integers(n))] for _ in range(3)])
            assert store.insert_rows(new) == n
            corpus = np.concatenate([corpus, new])
            dead = np.concatenate([dead, np.zeros(m, dtype=np.uint8)])
        elif op < 0.20:                                 # remove a few rows (some twice)
            for r in rng.integers(0, n, size=int(rng.integers(1, 6))):
                if not dead[r]:
                    store.remove_row(int(r))
                    dead[r] = 1
        elif op < 0.28:                                 # overwrite rows in place (a removed row stays removed)
            for r in rng.integers(0, n, size=int(rng.integers(1, 4))):
                if dead[r]:
                    assert L.trag_store_overwrite(store.handle, int(r), corpus[r].ctypes.data_as(fp), d) != 0
                    continue                            # a removed row cannot be written to
                vec = awkward_rows(rng, 1, d, [corpus[int(rng.integers(n))]])[0]
                st = L.trag_store_overwrite(store.handle, int(r), vec.ctypes.data_as(fp), d)
                assert st == 0, L.trag_last_error()
                corpus[r] = vec
        elif op < 0.34 and d % 8 == 0:                  # the bf16 copy comes and goes
            mirror = not mirror
            store.set_bf16_mirror(mirror)
            assert store.has_bf16_mirror() == mirror
        else:                                           # search
            live = int(n - dead.sum())
            batch = rng.random() < 0.4
            k = int(rng.choice([1, 3, 10, 50, 100, 257, 1000]))
            if not batch and rng.random() < 0.05:
                k = live + 5                            # more than there is (a global sort: single queries only)
            B = int(rng.choice([2, 3, 5, 9, 17, 40])) if batch else 1
            Q = rng.standard_normal((B, d)).astype(np.float32)
            u = rng.random()
            if u < 0.15:
                Q[0] = corpus[int(rng.integers(n))]     # a stored row: exact top, maybe a tie group
            elif u < 0.20:
                Q[0] = 0.0
            mode = int(rng.choice(modes_for(live, d, min(k, live), batch, mirror)))
            store.set_scan_mode(mode)
            what = f"walk {seed} step {step} mode {mode} k {k} B {B} n {n} live {live} mirror {mirror}"
            if batch:
                rows, scores, counts = store.search_batch_rows(Q, k)
                for b in range(B):
                    er, es = oracle.search(corpus, Q[b], k, metric, dead=dead)
                    assert counts[b] == len(er), what
                    assert_same_ranking(rows[b][:counts[b]], scores[b][:counts[b]], er, es, what + f" q{b}")
            else:
                rows, scores = store.search_rows(Q[0], k)
                er, es = oracle.search(corpus, Q[0], k, metric, dead=dead)
                assert_same_ranking(rows, scores, er, es, what)
            store.set_scan_mode(AUTO)
            searches += 1
    assert searches >= 40
    # the rows themselves survived the walk
    got = store.read_rows(0, corpus.shape[0])
    assert got.tobytes() == corpus.tobytes()
    store.close()
