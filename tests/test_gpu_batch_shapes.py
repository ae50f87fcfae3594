"""GPU: batches of seeded random shapes through the automatic routing (multi-query exact pass, f32 and bf16
matrix-core paths, tiled and query-stationary forms, with and without the bf16 mirror). Whatever route a
shape takes, rows and score bits are the oracle's (the reference's arithmetic, src/index.rs:394-462)."""
import os

import numpy as np
import pytest

from util import assert_same_ranking, unit_rows

pytestmark = pytest.mark.gpu

DIMS = [32, 64, 96, 128, 192, 256, 320, 384, 512, 768, 1024]


def draw_case(seed):
    rng = np.random.default_rng(seed)
    d = int(rng.choice(DIMS))
    n = int(rng.integers(3_000, 180_000 if d <= 384 else 90_000))
    B = int(rng.choice([2, 3, 5, 8, 9, 17, 33, 64, 100, 129, 200, 257, 300, 513, 700]))
    k = int(rng.choice([1, 3, 10, 37, 100, 150]))
    metric = int(rng.integers(0, 3))
    mirror = bool(rng.integers(0, 2)) and d % 8 == 0
    return rng, n, d, B, k, metric, mirror


@pytest.mark.parametrize("seed", list(range(40)))
def test_random_batch_shapes(trag, oracle, seed):
    # TRAG_WALK_SEED shifts the 40 shapes to fresh ones (for bug hunts; the default is what CI pins)
    rng, n, d, B, k, metric, mirror = draw_case(9000 + seed + 1000 * int(os.environ.get("TRAG_WALK_SEED", "0")))
    if metric == 0:
        corpus = unit_rows(rng, n, d)
    else:
        corpus = (rng.standard_normal((n, d)) * rng.choice([0.05, 1.0, 7.0])).astype(np.float32)
    # near-duplicates, an exact duplicate pair far apart, a zero row, rows on very different scales
    corpus[n // 2] = corpus[11]
    corpus[n // 3] = corpus[12] * np.float32(1.0 + 2.0 ** -12)
    corpus[7] = 0.0
    corpus[n - 1] *= np.float32(1e-4)
    Q = rng.standard_normal((B, d)).astype(np.float32)
    Q[0] = corpus[11]
    if B > 4:
        Q[4] = corpus[n // 3] + np.float32(1e-3) * rng.standard_normal(d).astype(np.float32)
    store = trag.VectorStore(trag.VectorStoreConfig(dimension=d, metric=trag.DistanceMetric(metric)))
    store.insert_rows(corpus)
    if mirror:
        store.set_bf16_mirror(True)
    dead = np.zeros(n, dtype=np.uint8)
    for r in {3, n // 2, int(rng.integers(0, n)), int(rng.integers(0, n))}:
        store.remove_row(r)
        dead[r] = 1
    rows, scores, cnt = store.search_batch_rows(Q, k)
    live = int(n - dead.sum())
    assert cnt.tolist() == [min(k, live)] * B
    what = f"seed {seed}: n={n} d={d} B={B} k={k} metric={metric} mirror={mirror} mode={store.last_stats()['mode_used']}"
    picks = sorted({0, 1, min(4, B - 1), B // 2, B - 1} | {int(x) for x in rng.integers(0, B, 3)})
    for b in picks:
        er, es = oracle.search(corpus, Q[b], k, metric, dead=dead)
        assert_same_ranking(rows[b][:cnt[b]], scores[b][:cnt[b]], er, es, f"{what} q={b}")
