"""CPU: trag_fuse (host C++ behind the C ABI) against the oracle and against the
reference's FusionStrategy unit tests (src/fusion.rs:242-616)."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
UNIT = json.load(open(os.path.join(HERE, "golden", "ref_unit_vectors.json")))
FUSE = json.load(open(os.path.join(HERE, "golden", "ref_fusion_vectors.json")))
CODE = {"RRF": 0, "Linear": 1, "Convex": 2, "DBSF": 3, "Union": 4, "Intersection": 5}


def strategy(trag, name, param):
    return trag.FusionStrategy(name, param)


@pytest.mark.parametrize("case", UNIT["rrf"], ids=lambda c: c["name"])
def test_rrf_reference_cases(trag, oracle, case):
    dense = [(i, 1.0) for i in case["dense"]]
    sparse = [(i, 1.0) for i in case["sparse"]]
    out = trag.FusionStrategy.RRF(case["k"]).fuse(dense, sparse)
    assert len(out) == case["expect_len"]
    if "expect_order" in case:
        assert [i for i, _ in out] == case["expect_order"]
    if "expect_first" in case:
        assert out[0][0] == case["expect_first"]
    if "expect_first_score" in case:
        assert abs(out[0][1] - case["expect_first_score"]) < case["tol"]
    ids, scores = oracle.rrf(case["dense"], case["sparse"], case["k"])
    assert [i for i, _ in out] == ids.tolist()
    assert np.array([s for _, s in out], dtype=np.float32).tobytes() == scores.tobytes()


@pytest.mark.parametrize("case", FUSE["cases"], ids=lambda c: c["name"])
def test_other_strategies_reference_cases(trag, oracle, case):
    dense = [(int(i), float(s)) for i, s in case["dense"]]
    sparse = [(int(i), float(s)) for i, s in case["sparse"]]
    out = strategy(trag, case["strategy"], case["param"]).fuse(dense, sparse)
    if "expect_len" in case:
        assert len(out) == case["expect_len"]
    if case.get("expect_nonempty"):
        assert out
    if "expect_first" in case:
        assert out[0][0] == case["expect_first"]
    if "expect_first_score" in case:
        assert abs(out[0][1] - case["expect_first_score"]) <= case["tol"]
    if "same_as" in case:
        other = strategy(trag, case["same_as"]["strategy"], case["same_as"]["param"]).fuse(dense, sparse)
        assert other == out
    ids, scores = oracle.fuse(CODE[case["strategy"]], case["param"], [i for i, _ in dense],
                              [s for _, s in dense], [i for i, _ in sparse], [s for _, s in sparse])
    assert [i for i, _ in out] == ids.tolist()
    assert np.array([s for _, s in out], dtype=np.float32).tobytes() == scores.tobytes()


def test_default_is_rrf_60(trag):
    f = trag.FusionStrategy.default()
    assert f.kind == "RRF" and abs(f.param - 60.0) < 0.01          # src/fusion.rs:245-251
    assert f.to_json() == {"RRF": {"k": 60.0}}
    lin = trag.FusionStrategy.from_json(trag.FusionStrategy.Linear(0.7).to_json())
    assert lin.kind == "Linear" and abs(lin.param - 0.7) < 0.01     # src/fusion.rs:254-265


@pytest.mark.parametrize("strategy_name", list(CODE))
def test_random_lists_match_oracle_bitwise(trag, oracle, strategy_name):
    """50+50 candidate lists as the pipeline produces (src/retrieve.rs:94,176),
    ~30% overlap, plus degenerate shapes."""
    rng = np.random.default_rng(hash(strategy_name) & 0xFFFF)
    for trial in range(40):
        nd, ns = int(rng.integers(0, 51)), int(rng.integers(0, 51))
        pool = rng.permutation(120)
        dense_ids = pool[:nd]
        sparse_ids = rng.permutation(np.concatenate([pool[:nd][: nd // 3], pool[60:60 + ns]]))[:ns]
        ds = np.sort(rng.standard_normal(nd).astype(np.float32))[::-1].copy()
        ss = np.sort(rng.random(ns).astype(np.float32) * 20)[::-1].copy()
        if trial % 7 == 0 and nd:
            ds[:] = ds[0]  # all-equal scores: min-max range 0, std 0 (src/fusion.rs:193-196,216-218)
        param = {"RRF": 60.0 if trial % 2 else 1.0, "Linear": 0.7, "Convex": 0.3}.get(strategy_name, 0.0)
        out = strategy(trag, strategy_name, param).fuse(list(zip(dense_ids.tolist(), ds.tolist())),
                                                        list(zip(sparse_ids.tolist(), ss.tolist())))
        ids, scores = oracle.fuse(CODE[strategy_name], param, dense_ids, ds, sparse_ids, ss)
        assert [i for i, _ in out] == ids.tolist()
        assert np.array([s for _, s in out], dtype=np.float32).tobytes() == scores.tobytes()
        if strategy_name == "RRF":
            assert all(s > 0 for _, s in out)


def test_intersection_subset_property(trag):
    """prop_intersection_subset, src/fusion.rs:580-597"""
    rng = np.random.default_rng(3)
    for _ in range(20):
        d = rng.permutation(30)[:10].tolist()
        s = rng.permutation(30)[:10].tolist()
        out = trag.FusionStrategy.Intersection().fuse([(i, 0.5) for i in d], [(i, 0.25) for i in s])
        assert {i for i, _ in out} == set(d) & set(s)


def test_merge_topk_matches_global_order(trag, oracle):
    import ctypes as C
    rng = np.random.default_rng(9)
    corpus = rng.standard_normal((4000, 16)).astype(np.float32)
    corpus[100] = corpus[3500]  # a cross-shard exact tie
    q = corpus[100].copy()
    k, shards = 10, 4
    per = 1000
    rows = np.zeros((shards, k), dtype=np.uint32)
    scores = np.zeros((shards, k), dtype=np.float32)
    n = np.zeros(shards, dtype=np.uint32)
    for s in range(shards):
        r, sc = oracle.search(corpus[s * per:(s + 1) * per], q, k, 0)
        rows[s, :len(r)] = r + s * per
        scores[s, :len(r)] = sc
        n[s] = len(r)
    out_r = np.zeros(k, dtype=np.uint32)
    out_s = np.zeros(k, dtype=np.float32)
    out_n = C.c_uint32()
    u32p, fp = C.POINTER(C.c_uint32), C.POINTER(C.c_float)
    st = trag.lib().trag_merge_topk(rows.ctypes.data_as(u32p), scores.ctypes.data_as(fp), n.ctypes.data_as(u32p),
                                    shards, k, k, out_r.ctypes.data_as(u32p), out_s.ctypes.data_as(fp),
                                    C.byref(out_n))
    assert st == 0 and out_n.value == k
    er, es = oracle.search(corpus, q, k, 0)
    assert out_r.tolist() == er.tolist() and out_s.tobytes() == es.tobytes()
