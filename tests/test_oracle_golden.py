"""CPU: the oracle against the reference's own known-answer tests (transcribed
in tests/golden/ref_unit_vectors.json) and against its committed fixtures."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "ref_unit_vectors.json")))
METRIC = {"cosine": 0, "euclidean": 1, "dot": 2}


def _case_rows(case):
    if case.get("rows_rule") == "one_hot_1000":
        rows = np.zeros((1000, 128), dtype=np.float32)
        rows[np.arange(1000), np.arange(1000) % 128] = 1.0
        return rows, np.ones(128, dtype=np.float32)
    return np.array(case["rows"], dtype=np.float32), np.array(case["query"], dtype=np.float32)


@pytest.mark.parametrize("case", GOLD["vector_store"], ids=lambda c: c["name"])
def test_vector_store_cases(oracle, case):
    rows, q = _case_rows(case)
    got_rows, got_scores = oracle.search(rows, q, case["k"], METRIC[case["metric"]])
    assert len(got_rows) == case["expect_len"]
    if "expect_order_prefix" in case:
        p = case["expect_order_prefix"]
        assert got_rows[:len(p)].tolist() == p
    if "expect_order_prefix_tie_rule" in case:
        p = case["expect_order_prefix_tie_rule"]
        assert got_rows[:len(p)].tolist() == p
    assert all(got_scores[i] >= got_scores[i + 1] for i in range(len(got_scores) - 1))
    # the reference-shaped variant (per-row vectors, full stable sort) agrees
    f_rows, f_scores = oracle.search_faithful(rows, q, case["k"], METRIC[case["metric"]])
    assert f_rows.tolist() == got_rows.tolist()
    assert f_scores.tobytes() == got_scores.tobytes()


def test_bench_shape_scores(oracle):
    case = [c for c in GOLD["vector_store"] if c["name"] == "bench_vector_search_shape"][0]
    rows, q = _case_rows(case)
    got_rows, got_scores = oracle.search(rows, q, 100, 0)
    assert got_rows.tolist() == list(range(100))
    expect = np.float32(1.0) / (np.sqrt(np.float32(128.0)) * np.float32(1.0))
    assert np.all(got_scores == expect)


@pytest.mark.parametrize("case", GOLD["scalar_fns"], ids=lambda c: c["fn"] + "@" + c["source"])
def test_scalar_fns(oracle, case):
    fn = {"cosine": oracle.cosine, "dot": oracle.dot, "euclidean": oracle.euclidean}[case["fn"]]
    assert abs(fn(case["a"], case["b"]) - case["expect"]) <= case["tol"]


@pytest.mark.parametrize("case", GOLD["rrf"], ids=lambda c: c["name"])
def test_rrf_cases(oracle, case):
    ids, scores = oracle.rrf(case["dense"], case["sparse"], case["k"])
    assert len(ids) == case["expect_len"]
    if "expect_order" in case:
        assert ids.tolist() == case["expect_order"]
    if "expect_first" in case:
        assert ids[0] == case["expect_first"]
    if "expect_first_score" in case:
        assert abs(scores[0] - case["expect_first_score"]) < case["tol"]
    assert np.all(scores > 0)  # prop_rrf_scores_positive, src/fusion.rs:548-565


def test_rrf_exact_value(oracle):
    # 1/(60+0+1) + 1/(60+0+1) in f32
    _, scores = oracle.rrf([1], [1], 60.0)
    one = np.float32(1.0) / np.float32(61.0)
    assert scores[0] == np.float32(one + one)


def test_sequential_sum_is_not_reassociated(oracle):
    """The oracle's dot is a strict left fold (src/index.rs:461): check against a
    Python-level f32 fold on data where pairwise summation differs."""
    rng = np.random.default_rng(5)
    a = rng.standard_normal(4096).astype(np.float32)
    b = rng.standard_normal(4096).astype(np.float32)
    s = np.float32(0.0)
    for x, y in zip(a, b):
        s = np.float32(s + np.float32(x * y))
    assert oracle.dot(a, b) == s
    na = np.float32(0.0)
    nb = np.float32(0.0)
    for x in a:
        na = np.float32(na + np.float32(x * x))
    for y in b:
        nb = np.float32(nb + np.float32(y * y))
    cos = np.float32(s / np.float32(np.sqrt(na) * np.sqrt(nb)))
    assert oracle.cosine(a, b) == cos


def test_generated_fixture_is_stable(oracle):
    """tests/golden/oracle_search_fixture.npz was produced by make_golden.py from
    this oracle; a change in the oracle's arithmetic or ordering shows up here."""
    fx = np.load(os.path.join(HERE, "golden", "oracle_search_fixture.npz"))
    for d in (3, 128, 384):
        corpus = oracle.synth_rows(int(fx[f"seed_{d}"]), 0, int(fx[f"n_{d}"]), d)
        queries = oracle.synth_rows(int(fx[f"seed_{d}"]) + 100, 0, 4, d)
        for m in range(3):
            for qi in range(4):
                rows, scores = oracle.search(corpus, queries[qi], 10, m)
                assert rows.tolist() == fx[f"rows_{d}_{m}"][qi].tolist()
                assert scores.tobytes() == fx[f"scores_{d}_{m}"][qi].tobytes()


def test_tie_and_nan_ordering(oracle):
    rows = np.array([[1, 0], [2, 0], [np.nan, 0], [1, 0], [0, 0]], dtype=np.float32)
    r, s = oracle.search(rows, np.array([1, 0], dtype=np.float32), 10, 0)
    # rows 0,1,3 score 1.0 (tie, ascending row), row 4 scores 0.0 (zero norm), NaN last
    assert r.tolist() == [0, 1, 3, 4, 2]
    assert np.isnan(s[-1])


def test_dead_rows_are_skipped(oracle):
    rng = np.random.default_rng(0)
    rows = rng.standard_normal((50, 8)).astype(np.float32)
    dead = np.zeros(50, dtype=np.uint8)
    dead[[3, 7, 9]] = 1
    r, _ = oracle.search(rows, rows[3], 50, 0, dead=dead)
    assert len(r) == 47 and not set(r.tolist()) & {3, 7, 9}
