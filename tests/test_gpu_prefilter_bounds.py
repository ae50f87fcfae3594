"""GPU: the bf16 prefilter's error bound is MEASURED (the relative rounding error of the query and
the largest one over the store's rows) and its thresholds tighten during the full pass. Both
must stay rigorous when the data is as unkind to bf16 as f32 values can be: the returned rows
and score bits are the oracle's in every case below (the reference's arithmetic,
src/index.rs:394-462)."""
import numpy as np
import pytest

from util import assert_same_ranking, unit_rows

pytestmark = pytest.mark.gpu

BF16, MFMA = 4, 3


def make_store(trag, dim, metric=0, mode=BF16, reserve=0):
    return trag.VectorStore(trag.VectorStoreConfig(dimension=dim, metric=trag.DistanceMetric(metric),
                                                   scan_mode=mode, reserve_rows=reserve))


def worst_for_bf16(x):
    """Every component moved just below the midpoint between two bf16 values: rounding to nearest
    then errs by (almost) the format's maximum, in the same direction for every component."""
    b = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    return ((b & np.uint32(0xFFFF0000)) | np.uint32(0x7FFF)).view(np.float32)


def check(store, oracle, corpus, Q, k, metric, what, dead=None, rows=None, modes=(BF16,)):
    got_rows, got_scores, cnt = store.search_batch_rows(Q, k)
    assert store.last_stats()["mode_used"] in modes
    for b in (range(len(Q)) if rows is None else rows):
        er, es = oracle.search(corpus, Q[b], k, metric, dead=dead)
        assert cnt[b] == len(er)
        assert_same_ranking(got_rows[b][:cnt[b]], got_scores[b][:cnt[b]], er, es, f"{what} q={b} k={k}")


@pytest.mark.parametrize("metric", [0, 2])
@pytest.mark.parametrize("d,B", [(384, 24), (384, 300), (768, 260)])
def test_rows_inserted_later_widen_the_bound(trag, oracle, metric, d, B):
    """A store of ordinary rows measures a rounding ratio near 0.4 * 2^-8. Rows whose every component
    rounds as badly as possible, all in one direction and parallel to a query, are inserted AFTERWARDS
    with exact scores on both sides of that query's k-th best: their approximate scores are off by the
    format's worst case, far more than the bound measured before them allowed. The inserts must have
    raised the store's ratio, or rows of the exact top k are lost."""
    rng = np.random.default_rng(1000 + d + B + metric)
    n = 70_000
    corpus = unit_rows(rng, n, d)
    Q = unit_rows(rng, B, d)
    store = make_store(trag, d, metric)
    store.insert_rows(corpus)
    check(store, oracle, corpus, Q, 10, metric, "before", rows=range(6))
    extra = []
    for b in range(4):
        er, es = oracle.search(corpus, Q[b], 10, 0)
        kth = float(es[-1])
        # cosine with Q[b] spread over [kth - 0.02, kth + 0.02]: mixtures of Q[b] and a random direction
        for c in np.linspace(kth - 0.02, kth + 0.02, 48):
            r = unit_rows(rng, 1, d)[0]
            r -= np.dot(r, Q[b]) * Q[b]
            r /= np.linalg.norm(r)
            v = (np.float32(c) * Q[b] + np.float32(np.sqrt(max(0.0, 1.0 - c * c))) * r).astype(np.float32)
            extra.append(worst_for_bf16(v))
    extra = np.stack(extra).astype(np.float32)
    store.insert_rows(extra)
    full = np.concatenate([corpus, extra])
    for k in (10, 50):
        check(store, oracle, full, Q, k, metric, "after", rows=list(range(8)) + [B - 1])


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_worst_case_queries_and_scales(trag, oracle, metric):
    """Queries that round as badly as possible (their own measured ratio is the format's worst case),
    rows spread over four orders of magnitude for the unnormalised metrics, and a threshold ladder
    that has to climb a long way: many near-duplicates of each query's best rows arrive late in
    the store."""
    rng = np.random.default_rng(77 + metric)
    n, d, B = 80_000, 384, 280
    corpus = unit_rows(rng, n, d) if metric == 0 else rng.standard_normal((n, d)).astype(np.float32)
    if metric != 0:
        corpus[: n // 4] *= np.float32(1e-3)
        corpus[n // 4: n // 2] *= np.float32(30.0)
    Q = worst_for_bf16(rng.standard_normal((B, d)).astype(np.float32))
    # the last 2000 rows: near-duplicates of the first queries (best rows seen last by the pass)
    for i in range(2000):
        b = i % 20
        corpus[n - 1 - i] = (Q[b] / np.linalg.norm(Q[b]) * np.float32(1.0 if metric == 0 else 2.0)
                             + np.float32(0.05 + 0.002 * (i // 20)) * rng.standard_normal(d)).astype(np.float32)
    store = make_store(trag, d, metric)
    store.insert_rows(corpus)
    dead = np.zeros(n, dtype=np.uint8)
    for r in (n - 1, n - 21, 7):
        store.remove_row(r)
        dead[r] = 1
    # (Euclidean distances over rows of such different lengths crowd within the bound of one another:
    # the lists may overflow for much of the batch, which is then repeated on the f32 matrix cores)
    for k in (1, 10, 100):
        check(store, oracle, corpus, Q, k, metric, f"metric {metric}", dead=dead,
              rows=list(range(24)) + list(range(B - 4, B)), modes=(BF16,) if metric != 1 else (BF16, MFMA))


def pow2_query(rng, d):
    """Components +-2^a: the query itself is exact in bf16, and a row whose magnitudes are proportional
    to its own meets the Cauchy-Schwarz worst case of the bound."""
    return (rng.choice([-1.0, 1.0], d) * np.exp2(rng.integers(-6, -2, d))).astype(np.float32)


def aligned_worst_row(rng, q, cos_target, up):
    """A row v with v_i = t_i q_i (1 + 2^-8 -+ 2^-23), t_i = +-1 chosen so that cos(q, v) ~ cos_target.
    Every component rounds to bf16 by the format's maximum, and in the direction that raises (`up`)
    or lowers the approximate dot product: it is off by 2^-8 ||q|| ||v||."""
    w = q.astype(np.float64) ** 2
    order = rng.permutation(len(q))
    plus = np.zeros(len(q), dtype=bool)
    want, have = (1.0 + cos_target) / 2.0 * w.sum(), 0.0
    for i in order[np.argsort(-w[order], kind="stable")]:   # large terms first, small ones settle the rest
        if have + w[i] <= want:
            plus[i] = True
            have += w[i]
    t = np.where(plus, 1.0, -1.0).astype(np.float32)
    bits = np.abs(q).view(np.uint32).copy()                 # magnitude 2^a: mantissa 0
    grow = plus if up else ~plus                            # magnitudes that must round upwards
    bits |= np.where(grow, np.uint32(0x8001), np.uint32(0x7FFF))
    return (t * np.sign(q) * bits.view(np.float32)).astype(np.float32)


def exact_in_bf16(x):
    """The nearest values that bf16 holds exactly (rows made of them round without error)."""
    b = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    return ((b + np.uint32(0x8000)) & np.uint32(0xFFFF0000)).view(np.float32)


@pytest.mark.parametrize("d,B", [(384, 24), (384, 300), (256, 140), (768, 70)])
def test_rows_at_the_formats_worst_case_move_the_bound(trag, oracle, d, B):
    """The adversary the measured bound must survive. The store first holds rows that bf16 represents
    exactly (measured ratio 0: the bound is the f32 summation terms alone, some 2e-4) and the queries
    are exact in bf16 too. Then come rows whose exact scores lie just ABOVE a query's k-th best but
    whose approximate scores are too low by the full 2^-8 ||q|| ||v|| -- twenty times the bound
    measured before they arrived: unless their insertion raised the store's ratio they fall below the
    candidate threshold, nothing left in the list contradicts the final check, and rows of the exact
    top k are lost."""
    rng = np.random.default_rng(4000 + d + B)
    n, k = 70_000, 10
    corpus = exact_in_bf16(unit_rows(rng, n, d))
    Q = exact_in_bf16(unit_rows(rng, B, d))
    Q[0] = pow2_query(rng, d)
    Q[1] = pow2_query(rng, d)
    store = make_store(trag, d, 0)
    store.insert_rows(corpus)
    check(store, oracle, corpus, Q, k, 0, "exact-in-bf16 rows", rows=[0, 1, 2])
    extra = []
    for b in (0, 1):
        er, es = oracle.search(corpus, Q[b], k, 0)
        kth = float(es[-1])
        for j in range(1, 13):
            extra.append(aligned_worst_row(rng, Q[b], kth + 0.00025 * j, up=False))
    extra = np.stack(extra).astype(np.float32)
    # one at a time: the deferred small inserts must keep the store's ratio up to date as well
    for row in extra:
        store.insert_rows(row[None, :])
    full = np.concatenate([corpus, extra])
    for kk in (k, 3):
        check(store, oracle, full, Q, kk, 0, "worst-case rows", rows=[0, 1, 2, 3, B - 1])


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_long_candidate_lists_are_pruned_exactly(trag, oracle, metric):
    """Many wanted rows per query and a batch of only a few workgroups: the candidate lists of the
    full pass run past 512 entries (printed by a TRAG_LIST_DIAG build: 530-800 here),
    the regime where the prune stage selects its threshold by radix levels and ranks only the
    survivors. Rows and score bits must still be the oracle's, for every k."""
    rng = np.random.default_rng(31 + metric)
    n, d, B = 300_000, 128, 160
    corpus = unit_rows(rng, n, d) if metric != 2 else rng.standard_normal((n, d)).astype(np.float32)
    Q = unit_rows(rng, B, d)
    store = make_store(trag, d, metric)
    store.insert_rows(corpus)
    for k in (100, 200):
        check(store, oracle, corpus, Q, k, metric, f"long lists metric {metric}", rows=[0, 1, 2, 77, B - 1],
              modes=(BF16, MFMA))
