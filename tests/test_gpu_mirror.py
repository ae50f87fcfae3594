"""GPU: searches that read the optional bf16 mirror of the corpus must return the
reference's f32 results bit for bit (rows, ranks, score bits), like every other mode."""
import numpy as np
import pytest

from util import assert_same_ranking, unit_rows

pytestmark = pytest.mark.gpu

EXACT, BF16, MIRROR = 1, 4, 5


def make_store(trag, dim, metric=0, mode=0, reserve=0):
    return trag.VectorStore(trag.VectorStoreConfig(dimension=dim, metric=trag.DistanceMetric(metric),
                                                   scan_mode=mode, reserve_rows=reserve))


def adversarial_corpus(rng, n, d, metric):
    corpus = unit_rows(rng, n, d) if metric == 0 else rng.standard_normal((n, d)).astype(np.float32)
    corpus[17] = corpus[n - 5]            # duplicate rows in different tiles
    corpus[1000] = 0.0                    # zero-norm row
    corpus[2000] *= 1e-3                  # a short row
    corpus[3000] *= np.float32(1e-30)     # squared norm below the bound's range: always scored exactly
    # every element just below a bf16 rounding boundary: the worst case of the mirror's rounding
    corpus[4000] = np.float32(1.0 + 2.0 ** -8 - 2.0 ** -20) * np.sign(corpus[4000]) / np.float32(np.sqrt(d))
    return np.ascontiguousarray(corpus, dtype=np.float32)


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("n,d", [(70_000, 384), (90_000, 768), (66_000, 128), (66_000, 4096), (100_000, 256)])
def test_mirror_single_query_parity(trag, oracle, metric, n, d):
    rng = np.random.default_rng(n + d + metric)
    corpus = adversarial_corpus(rng, n, d, metric)
    store = make_store(trag, d, metric, MIRROR)
    store.set_bf16_mirror(True)                      # before the rows: insert keeps it in step
    assert store.has_bf16_mirror()
    store.insert_rows(corpus[: n // 2])
    store.insert_rows(corpus[n // 2:])
    dead = np.zeros(n, dtype=np.uint8)
    for r in (5, 64, 12_345, n - 5):
        store.remove_row(r)
        dead[r] = 1
    Q = rng.standard_normal((8, d)).astype(np.float32)
    Q[1] = corpus[17]                                # exact tie at the top (row n-5 is removed: row 17 alone)
    Q[2] = corpus[4242] * np.float32(1.0001) + np.float32(1e-4)   # a near-duplicate
    Q[3] = corpus[4000]                              # the rounding worst case
    Q[4] = 0.0                                       # zero query: cosine ties at 0.0 everywhere -> exact repeat
    fallbacks = 0
    for b in range(8):
        for k in (1, 10, 100):
            rows, scores = store.search_rows(Q[b], k)
            er, es = oracle.search(corpus, Q[b], k, metric, dead=dead)
            assert_same_ranking(rows, scores, er, es, f"mirror n={n} d={d} q={b} k={k} metric={metric}")
            st = store.last_stats()
            assert st["mode_used"] == MIRROR
            fallbacks += st["fallbacks"]
    # random queries on a random corpus are served by the mirror itself
    assert fallbacks <= 3 * 4, fallbacks


def test_mirror_follows_insert_overwrite_and_growth(trag, oracle):
    n, d = 70_000, 384
    rng = np.random.default_rng(9)
    corpus = unit_rows(rng, n, d)
    store = make_store(trag, d, 0, MIRROR)
    store.insert_rows(corpus[:66_000])
    store.set_bf16_mirror(True)                      # built from the rows present
    q = unit_rows(rng, 1, d)[0]
    rows, scores = store.search_rows(q, 10)
    er, es = oracle.search(corpus[:66_000], q, 10, 0)
    assert_same_ranking(rows, scores, er, es, "mirror built after insert")
    assert store.last_stats()["mode_used"] == MIRROR and store.last_stats()["fallbacks"] == 0
    store.insert_rows(corpus[66_000:])               # grows the store: the mirror moves with it
    # a new best row through overwrite: the mirror must see it
    c = trag.Chunk(content="x", embedding=list(corpus[123]))
    store2 = make_store(trag, d, 0, 0)
    store2.set_bf16_mirror(True)
    store2.insert_rows(corpus)
    store2.insert(c)
    c.embedding = list(q)
    store2.insert(c)                                 # same ChunkId: overwrite of row n
    rows2, scores2 = store2.search_rows(q, 10)
    full = np.concatenate([corpus, q[None, :]])
    er2, es2 = oracle.search(full, q, 10, 0)
    assert_same_ranking(rows2, scores2, er2, es2, "mirror after overwrite")
    assert rows2[0] == n and store2.last_stats()["mode_used"] == MIRROR  # AUTO uses the mirror
    rows, scores = store.search_rows(q, 10)
    er, es = oracle.search(corpus, q, 10, 0)
    assert_same_ranking(rows, scores, er, es, "mirror after growth")
    store.set_bf16_mirror(False)
    assert not store.has_bf16_mirror()
    rows3, scores3 = store.search_rows(q, 10)
    assert store.last_stats()["mode_used"] == EXACT
    assert np.array_equal(rows3, rows) and scores3.tobytes() == scores.tobytes()


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("n,d,B", [(70_000, 384, 9), (100_000, 768, 40), (80_000, 128, 200), (131_072, 256, 300)])
def test_mirror_batch_parity(trag, oracle, metric, n, d, B):
    rng = np.random.default_rng(n + d + B)
    corpus = adversarial_corpus(rng, n, d, metric)
    Q = rng.standard_normal((B, d)).astype(np.float32)
    Q[3] = corpus[17]
    Q[4] = corpus[4242] * np.float32(1.0001) + np.float32(1e-4)
    Q[5] = 0.0
    store = make_store(trag, d, metric, MIRROR)
    store.insert_rows(corpus)
    store.set_bf16_mirror(True)
    dead = np.zeros(n, dtype=np.uint8)
    for r in (5, 64, 12_345, n - 5):
        store.remove_row(r)
        dead[r] = 1
    plain = make_store(trag, d, metric, BF16)        # the same pass converting the f32 corpus on the fly
    plain.insert_rows(corpus)
    for r in (5, 64, 12_345, n - 5):
        plain.remove_row(r)
    for k in (10, 100):
        rows, scores, cnt = store.search_batch_rows(Q, k)
        assert cnt.tolist() == [k] * B
        assert store.last_stats()["mode_used"] == BF16   # batches: the matrix-core prefilter on the mirror
        check = range(B) if B <= 40 else list(range(8)) + list(range(B - 8, B))
        for b in check:
            er, es = oracle.search(corpus, Q[b], k, metric, dead=dead)
            assert_same_ranking(rows[b], scores[b], er, es, f"mirror batch n={n} d={d} q={b} k={k} metric={metric}")
        prow, pscore, _ = plain.search_batch_rows(Q, k)
        assert np.array_equal(prow, rows) and pscore.tobytes() == scores.tobytes()


def test_mirror_needs_dim_multiple_of_8(trag):
    store = make_store(trag, 12)
    with pytest.raises(trag.TragError):
        store.set_bf16_mirror(True)
    assert not store.has_bf16_mirror()
