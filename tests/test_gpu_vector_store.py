"""GPU: the HIP path through the C ABI against the CPU oracle (bit-exact rows,
ranks and score bits) and against the reference's VectorStore unit tests
(src/index.rs:667-936)."""
import json
import os

import numpy as np
import pytest

from util import assert_same_ranking, unit_rows

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "ref_unit_vectors.json")))
METRIC = {"cosine": 0, "euclidean": 1, "dot": 2}
EXACT, STREAM = 1, 2


def make_store(trag, dim, metric=0, mode=0, reserve=0):
    return trag.VectorStore(trag.VectorStoreConfig(dimension=dim, metric=trag.DistanceMetric(metric),
                                                   scan_mode=mode, reserve_rows=reserve))


def chunk(trag, emb, text="x"):
    return trag.Chunk(content=text, embedding=None if emb is None else list(emb))


def test_single_hip_runtime(trag):
    """libtrag.so and torch must share one HIP runtime, or device pointers
    cannot cross between them."""
    import torch
    from trueno_rag_amd import _abi
    trag.lib()
    assert torch.cuda.is_available()
    assert len(_abi.hip_runtimes_loaded()) == 1, _abi.hip_runtimes_loaded()


# ---------------- the reference's own unit tests ----------------

def test_vector_store_new(trag):
    store = trag.VectorStore.with_dimension(384)
    assert store.config().dimension == 384 and store.is_empty()


def test_vector_store_insert_get_remove(trag):
    store = trag.VectorStore.with_dimension(3)
    c = chunk(trag, [1.0, 0.0, 0.0])
    store.insert(c)
    assert store.len() == 1 and not store.is_empty() and store.get(c.id) is c
    assert store.remove(c.id) is c
    assert store.len() == 0 and store.get(c.id) is None
    assert store.remove(trag.ChunkId()) is None                 # src/index.rs:815-819
    assert store.search([1.0, 0.0, 0.0], 5) == []


def test_insert_errors(trag):
    store = trag.VectorStore.with_dimension(3)
    with pytest.raises(trag.InvalidConfig):                      # src/index.rs:704-710
        store.insert(chunk(trag, None))
    with pytest.raises(trag.DimensionMismatch) as e:             # src/index.rs:713-726
        store.insert(chunk(trag, [1.0, 0.0]))
    assert (e.value.expected, e.value.actual) == (3, 2)
    with pytest.raises(trag.DimensionMismatch) as e:             # src/index.rs:793-797
        store.search([1.0, 0.0], 10)
    assert (e.value.expected, e.value.actual) == (3, 2)
    assert store.len() == 0


def test_c_abi_error_codes(trag):
    import ctypes as C
    L = trag.lib()
    store = trag.VectorStore.with_dimension(3)
    v = np.zeros(2, dtype=np.float32)
    fp = C.POINTER(C.c_float)
    assert L.trag_store_insert(store.handle, v.ctypes.data_as(fp), 1, 2, None) == 1
    assert b"expected 3" in L.trag_last_error()
    assert L.trag_store_insert(store.handle, None, 1, 3, None) == 2
    n = C.c_uint32(77)
    assert L.trag_store_search(store.handle, v.ctypes.data_as(fp), 2, 5, None, None, C.byref(n)) == 1
    assert L.trag_store_remove(store.handle, 0) == 5


@pytest.mark.parametrize("case", GOLD["vector_store"], ids=lambda c: c["name"])
def test_reference_search_cases(trag, oracle, case):
    if case.get("rows_rule") == "one_hot_1000":
        rows = np.zeros((1000, 128), dtype=np.float32)
        rows[np.arange(1000), np.arange(1000) % 128] = 1.0
        q = np.ones(128, dtype=np.float32)
    else:
        rows, q = np.array(case["rows"], dtype=np.float32), np.array(case["query"], dtype=np.float32)
    for mode in (EXACT, STREAM):
        store = make_store(trag, rows.shape[1], METRIC[case["metric"]], mode)
        chunks = [chunk(trag, r, f"chunk {i}") for i, r in enumerate(rows)]
        store.insert_batch(chunks)
        res = store.search(q, case["k"])
        assert len(res) == case["expect_len"]
        order = [chunks.index(store.get(cid)) for cid, _ in res]
        for key in ("expect_order_prefix", "expect_order_prefix_tie_rule"):
            if key in case:
                assert order[:len(case[key])] == case[key]
        er, es = oracle.search(rows, q, case["k"], METRIC[case["metric"]])
        assert_same_ranking(order, [s for _, s in res], er, es, case["name"])
        if case.get("rows_rule"):                                 # the bench also asks for top-100
            r100, s100 = store.search_rows(q, 100)
            er, es = oracle.search(rows, q, 100, 0)
            assert_same_ranking(r100, s100, er, es, "search_top_100")


def test_prop_search_returns_stored(trag):
    """prop_vector_store_search_returns_stored, src/index.rs:913-935"""
    rng = np.random.default_rng(1)
    for _ in range(8):
        dim, n = int(rng.integers(2, 10)), int(rng.integers(1, 20))
        store = trag.VectorStore.with_dimension(dim)
        ids = []
        for i in range(n):
            e = [0.0] * dim
            e[i % dim] = 1.0
            c = chunk(trag, e, f"chunk {i}")
            ids.append(c.id)
            store.insert(c)
        res = store.search([1.0] * dim, n)
        assert len(res) == n and all(cid in ids for cid, _ in res)


# ---------------- parity with the oracle on seeded inputs ----------------

SHAPES = [(1, 3), (5, 2), (63, 8), (64, 100), (65, 128), (700, 36), (1000, 384), (3000, 768), (300, 4096),
          (2500, 256), (999, 7)]


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("n,d", SHAPES)
def test_parity_small(trag, oracle, n, d, metric):
    rng = np.random.default_rng(n * 131 + d)
    corpus = unit_rows(rng, n, d) if metric == 0 else rng.standard_normal((n, d)).astype(np.float32)
    queries = rng.standard_normal((3, d)).astype(np.float32)
    for mode in (EXACT, STREAM):
        store = make_store(trag, d, metric, mode)
        store.insert_rows(corpus)
        assert store.len() == n
        for q in queries:
            for k in (1, 10, 50, 100):
                rows, scores = store.search_rows(q, k)
                er, es = oracle.search(corpus, q, k, metric)
                assert_same_ranking(rows, scores, er, es, f"n={n} d={d} metric={metric} mode={mode} k={k}")


@pytest.mark.parametrize("mode", [EXACT, STREAM])
def test_parity_100k_x_384(trag, oracle, mode):
    n, d = 100_000, 384
    store = make_store(trag, d, 0, mode, reserve=n)
    store.fill_synthetic(n, 1002)
    corpus = oracle.synth_rows(1002, 0, n, d)
    got = store.read_rows(0, n)
    assert got.tobytes() == corpus.tobytes(), "device generator and host generator differ"
    queries = oracle.synth_rows(1102, 0, 8, d)
    for q in queries:
        for k in (10, 50, 100):
            rows, scores = store.search_rows(q, k)
            er, es = oracle.search(corpus, q, k, 0)
            assert_same_ranking(rows, scores, er, es, f"100k mode={mode} k={k}")
    st = store.last_stats()
    assert st["fallbacks"] == 0
    assert st["mode_used"] == mode


def test_batch_matches_single(trag, oracle):
    n, d, B, k = 20_000, 128, 17, 10
    rng = np.random.default_rng(77)
    corpus = unit_rows(rng, n, d)
    Q = unit_rows(rng, B, d)
    store = make_store(trag, d)
    store.insert_rows(corpus)
    rows, scores, cnt = store.search_batch_rows(Q, k)
    assert cnt.tolist() == [k] * B
    for b in range(B):
        er, es = oracle.search(corpus, Q[b], k, 0)
        assert_same_ranking(rows[b], scores[b], er, es, f"batch query {b}")


def test_device_pointer_entry(trag, oracle):
    import ctypes as C
    import torch
    n, d, B, k = 30_000, 384, 4, 10
    store = make_store(trag, d, 0, EXACT, reserve=n)
    store.fill_synthetic(n, 5)
    corpus = store.read_rows(0, n)
    Q = oracle.synth_rows(6, 0, B, d)
    dQ = torch.from_numpy(Q).cuda()
    d_rows = torch.zeros((B, k), dtype=torch.int32, device="cuda")
    d_scores = torch.zeros((B, k), dtype=torch.float32, device="cuda")
    d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
    d_st = torch.ones(B, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream()
    rc = trag.lib().trag_store_search_device(store.handle, dQ.data_ptr(), B, d, k, d_rows.data_ptr(),
                                             d_scores.data_ptr(), d_n.data_ptr(), d_st.data_ptr(),
                                             C.c_void_p(stream.cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    assert d_n.tolist() == [k] * B and d_st.tolist() == [0] * B
    for b in range(B):
        er, es = oracle.search(corpus, Q[b], k, 0)
        assert_same_ranking(d_rows[b].cpu().numpy().astype(np.uint32), d_scores[b].cpu().numpy(), er, es,
                            f"device entry query {b}")


# ---------------- edge cases ----------------

@pytest.mark.parametrize("mode", [EXACT, STREAM])
def test_k_edges_and_empty(trag, oracle, mode):
    d = 128
    store = make_store(trag, d, 0, mode)
    q = np.ones(d, dtype=np.float32)
    assert store.search_rows(q, 10)[0].size == 0            # empty store (src/index.rs:409)
    rng = np.random.default_rng(2)
    corpus = unit_rows(rng, 37, d)
    store.insert_rows(corpus)
    assert store.search_rows(q, 0)[0].size == 0             # k == 0
    rows, scores = store.search_rows(q, 1000)               # k > len: all rows, sorted
    er, es = oracle.search(corpus, q, 1000, 0)
    assert rows.size == 37
    assert_same_ranking(rows, scores, er, es, "k>n")


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("mode", [EXACT, STREAM])
def test_zero_nan_inf_and_duplicates(trag, oracle, mode, metric):
    d = 128
    rng = np.random.default_rng(3)
    corpus = rng.standard_normal((500, d)).astype(np.float32)
    corpus[7] = 0.0                       # zero-norm row -> cosine exactly 0.0 (src/index.rs:445-446)
    corpus[11] = corpus[400]              # exact duplicates: tie broken by lower row
    corpus[12] = corpus[400]
    corpus[20, 5] = np.nan                # NaN row ranks last
    corpus[30] *= 1e-20                   # tiny norm
    corpus[31] *= 1e15                    # large norm
    store = make_store(trag, d, metric, mode)
    store.insert_rows(corpus)
    for q in (corpus[400].copy(), np.zeros(d, dtype=np.float32), corpus[7] + 1.0):
        for k in (3, 10, 500):
            rows, scores = store.search_rows(q, k)
            er, es = oracle.search(corpus, q, k, metric)
            assert_same_ranking(rows, scores, er, es, f"edge metric={metric} mode={mode} k={k}")


@pytest.mark.parametrize("mode", [EXACT, STREAM])
def test_remove_and_overwrite(trag, oracle, mode):
    d, n = 384, 2000
    rng = np.random.default_rng(4)
    corpus = unit_rows(rng, n, d)
    store = make_store(trag, d, 0, mode)
    store.insert_rows(corpus)
    q = corpus[123].copy()
    dead = np.zeros(n, dtype=np.uint8)
    for r in (123, 0, 1999, 64, 65):
        store.remove_row(r)
        dead[r] = 1
    assert store.len() == n - 5
    newv = unit_rows(rng, 1, d)[0]
    import ctypes as C
    assert trag.lib().trag_store_overwrite(store.handle, 77, newv.ctypes.data_as(C.POINTER(C.c_float)), d) == 0
    corpus[77] = newv
    for k in (10, 100):
        rows, scores = store.search_rows(q, k)
        er, es = oracle.search(corpus, q, k, 0, dead=dead)
        assert_same_ranking(rows, scores, er, es, f"remove mode={mode}")
    assert 123 not in store.search_rows(q, n)[0]
    # duplicate ChunkId overwrites (src/index.rs:372)
    s2 = trag.VectorStore.with_dimension(3)
    c = chunk(trag, [1.0, 0.0, 0.0])
    s2.insert(c)
    c2 = trag.Chunk(content="again", embedding=[0.0, 1.0, 0.0], id=c.id)
    s2.insert(c2)
    assert s2.len() == 1 and s2.search([0.0, 1.0, 0.0], 1)[0][1] == 1.0


@pytest.mark.parametrize("mode", [EXACT, STREAM])
def test_huge_tie_group_is_cut_by_row(trag, oracle, mode):
    """More rows tie on the k-th score than the final sort can hold: the lowest
    rows win (bench-shaped one-hot data, 20 000 rows)."""
    d, n = 128, 20_000
    rows = np.zeros((n, d), dtype=np.float32)
    rows[np.arange(n), np.arange(n) % d] = 1.0
    rows[15_000] *= 3.0          # same cosine, different dot
    rows[17_000, :] = 1.0        # the single best row
    store = make_store(trag, d, 0, mode)
    store.insert_rows(rows)
    q = np.ones(d, dtype=np.float32)
    for k in (10, 100):
        got_r, got_s = store.search_rows(q, k)
        er, es = oracle.search(rows, q, k, 0)
        assert_same_ranking(got_r, got_s, er, es, f"tie cut mode={mode}")
    assert store.last_stats()["refine_levels"] == 3


@pytest.mark.parametrize("mode", [EXACT, STREAM])
def test_clustered_scores_need_refinement(trag, oracle, mode):
    """Scores packed into a narrow band (one coarse bin holds far more than the
    final sort's capacity) force the second radix level."""
    d, n = 384, 60_000
    rng = np.random.default_rng(8)
    centre = unit_rows(rng, 1, d)[0]
    corpus = centre[None, :] + 0.02 * rng.standard_normal((n, d)).astype(np.float32)
    corpus = np.ascontiguousarray(corpus, dtype=np.float32)
    store = make_store(trag, d, 0, mode)
    store.insert_rows(corpus)
    q = centre + 0.02 * rng.standard_normal(d).astype(np.float32)
    for k in (10, 100):
        rows, scores = store.search_rows(q, k)
        er, es = oracle.search(corpus, q, k, 0)
        assert_same_ranking(rows, scores, er, es, f"clustered mode={mode}")
    assert store.last_stats()["refine_levels"] >= 2


def test_growth_keeps_rows(trag, oracle):
    d = 128
    rng = np.random.default_rng(5)
    store = make_store(trag, d)
    parts = [unit_rows(rng, n, d) for n in (10, 1500, 3000, 1)]
    for p in parts:
        store.insert_rows(p)
    corpus = np.concatenate(parts)
    assert store.read_rows(0, corpus.shape[0]).tobytes() == corpus.tobytes()
    q = corpus[2000]
    rows, scores = store.search_rows(q, 10)
    er, es = oracle.search(corpus, q, 10, 0)
    assert_same_ranking(rows, scores, er, es, "growth")


def test_hybrid_retriever_path(trag, oracle):
    """src/retrieve.rs:175-220 with a stub embedder and a stub sparse index."""
    d, n = 128, 300
    rng = np.random.default_rng(6)
    corpus = unit_rows(rng, n, d)

    class Emb:
        def embed_query(self, text):
            if not text:
                raise ValueError("empty query")
            return corpus[int(text)]

    class Sparse:
        def __init__(self):
            self.ids = []

        def add(self, c):
            self.ids.append(c.id)

        def search(self, query, k):
            base = int(query)
            return [(self.ids[(base + 7 * i) % len(self.ids)], 10.0 - i) for i in range(min(k, 50))]

    hr = trag.HybridRetriever.new(trag.VectorStore.with_dimension(d), Sparse(), Emb())
    chunks = [chunk(trag, corpus[i], f"doc {i}") for i in range(n)]
    hr.index_batch(chunks)
    assert hr.len() == n
    res = hr.retrieve("42", 10)
    assert len(res) == 10 and res[0].chunk is chunks[42]          # best in both lists
    assert all(r.fused_score is not None for r in res)
    assert all(res[i].fused_score >= res[i + 1].fused_score for i in range(9))
    # same fusion by hand from the oracle
    er, es = oracle.search(corpus, corpus[42], 50, 0)
    sparse_rows = [(42 + 7 * i) % n for i in range(50)]
    ids, fs = oracle.rrf(er.astype(np.uint64), np.array(sparse_rows, dtype=np.uint64), 60.0)
    assert [chunks.index(r.chunk) for r in res] == ids[:10].tolist()
    assert np.array([r.fused_score for r in res], dtype=np.float32).tobytes() == fs[:10].tobytes()
    dres = hr.retrieve_dense("42", 5)
    assert [chunks.index(r.chunk) for r in dres] == er[:5].tolist()
    dr = trag.DenseRetriever.new(hr.dense_store(), Emb())
    assert [chunks.index(r.chunk) for r in dr.retrieve("42", 5)] == er[:5].tolist()


# ---------------- batched queries on the matrix cores ----------------

MFMA = 3


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("n,d,B", [(50_000, 768, 40), (33_333, 384, 130), (20_000, 128, 300), (30_000, 256, 9),
                                   (25_001, 384, 32), (20_000, 64, 64),
                                   # an odd number of 32-column chunks: the exact re-score's last, unpaired chunk
                                   (20_000, 96, 20), (20_000, 160, 12), (20_000, 32, 10)])
def test_batch_mfma_parity(trag, oracle, n, d, B, metric):
    rng = np.random.default_rng(n + d + B)
    corpus = unit_rows(rng, n, d) if metric == 0 else rng.standard_normal((n, d)).astype(np.float32)
    corpus[17] = corpus[n - 5]            # duplicate rows across tiles
    corpus[1000] = 0.0                    # zero-norm row
    Q = rng.standard_normal((B, d)).astype(np.float32)
    Q[3] = corpus[17]                     # a query with an exact tie at the top
    Q[5] = 0.0                            # zero query
    store = make_store(trag, d, metric, MFMA)
    store.insert_rows(corpus)
    dead = np.zeros(n, dtype=np.uint8)
    for r in (5, 64, 12_345):
        store.remove_row(r)
        dead[r] = 1
    for k in (10, 50):
        rows, scores, cnt = store.search_batch_rows(Q, k)
        assert cnt.tolist() == [k] * B
        for b in range(B):
            er, es = oracle.search(corpus, Q[b], k, metric, dead=dead)
            assert_same_ranking(rows[b], scores[b], er, es, f"mfma batch n={n} d={d} q={b} k={k} metric={metric}")
    st = store.last_stats()
    assert st["mode_used"] == MFMA
    # the zero query ties every row at 0.0: it cannot be certified and is repeated exactly
    assert st["fallbacks"] <= 2


def test_batch_auto_mode_uses_mfma_above_8_queries(trag, oracle):
    n, d = 30_000, 384
    store = make_store(trag, d, 0, 0, reserve=n)
    store.fill_synthetic(n, 77)
    corpus = store.read_rows(0, n)
    Q = oracle.synth_rows(78, 0, 17, d)
    rows, scores, cnt = store.search_batch_rows(Q, 10)
    assert store.last_stats()["mode_used"] == MFMA
    for b in range(17):
        er, es = oracle.search(corpus, Q[b], 10, 0)
        assert_same_ranking(rows[b], scores[b], er, es, f"auto batch {b}")
    rows, scores, cnt = store.search_batch_rows(Q[:9], 10)
    assert store.last_stats()["mode_used"] == MFMA
    rows, scores, cnt = store.search_batch_rows(Q[:8], 10)
    assert store.last_stats()["mode_used"] == EXACT
    for b in range(8):
        er, es = oracle.search(corpus, Q[b], 10, 0)
        assert_same_ranking(rows[b], scores[b], er, es, f"auto multi {b}")


def test_batch_euclidean_on_matrix_cores(trag, oracle):
    n, d, B = 8000, 128, 12
    rng = np.random.default_rng(12)
    corpus = rng.standard_normal((n, d)).astype(np.float32)
    corpus[100] = corpus[7000]                 # duplicates: distance ties
    corpus[4000] *= 1e-3                       # a short row
    Q = rng.standard_normal((B, d)).astype(np.float32)
    Q[2] = corpus[100]                         # distance 0 to two rows
    Q[3] = corpus[55] + np.float32(1e-4)       # a near-duplicate: cancellation in the squared form
    store = make_store(trag, d, 1, 0)
    store.insert_rows(corpus)
    rows, scores, _ = store.search_batch_rows(Q, 10)
    assert store.last_stats()["mode_used"] == MFMA
    for b in range(B):
        er, es = oracle.search(corpus, Q[b], 10, 1)
        assert_same_ranking(rows[b], scores[b], er, es, f"euclid batch {b}")
    assert rows[2][:2].tolist() == [100, 7000]
    # unit-norm rows at 384 dims, a larger corpus: nearest-neighbour gaps are small there
    n2, d2 = 60_000, 384
    store2 = make_store(trag, d2, 1, 0, reserve=n2)
    store2.fill_synthetic(n2, 31)
    corpus2 = store2.read_rows(0, n2)
    Q2 = oracle.synth_rows(32, 0, 20, d2)
    rows, scores, _ = store2.search_batch_rows(Q2, 10)
    for b in range(20):
        er, es = oracle.search(corpus2, Q2[b], 10, 1)
        assert_same_ranking(rows[b], scores[b], er, es, f"euclid unit batch {b}")
    assert store2.last_stats()["fallbacks"] <= 2


# ---------------- cross-shard exchange kernels (one GPU stands in for two shards) ----------------

def test_two_shards_on_one_gpu_merge_to_global_topk(trag, oracle):
    import ctypes as C
    import torch
    from trueno_rag_amd.sharded import ShardExchange, shard_range
    n_total, d, B, k = 40_001, 384, 5, 10
    corpus = oracle.synth_rows(9001, 0, n_total, d)
    corpus[39_000] = corpus[10]                       # exact tie across the shard boundary
    Q = np.ascontiguousarray(np.concatenate([oracle.synth_rows(9002, 0, B - 1, d), corpus[10:11]]))
    rec = 2 * k + 1
    gathered = torch.zeros((2, B, rec), dtype=torch.int32, device="cuda")
    L = trag.lib()
    stores = []
    for rank in range(2):
        row0, n_local = shard_range(rank, 2, n_total)
        st = make_store(trag, d, 0, EXACT)
        st.insert_rows(corpus[row0:row0 + n_local])
        stores.append(st)
        rows, scores, cnt = st.search_batch_rows(Q, k)
        d_rows = torch.from_numpy(rows.astype(np.int32)).cuda()
        d_scores = torch.from_numpy(scores).cuda()
        d_n = torch.from_numpy(cnt.astype(np.int32)).cuda()
        assert L.trag_pack_topk_device(C.c_void_p(d_rows.data_ptr()), C.c_void_p(d_scores.data_ptr()),
                                       C.c_void_p(d_n.data_ptr()), row0, B, k,
                                       C.c_void_p(gathered[rank].data_ptr()), None) == 0
    out_r = torch.zeros((B, k), dtype=torch.int32, device="cuda")
    out_s = torch.zeros((B, k), dtype=torch.float32, device="cuda")
    out_n = torch.zeros(B, dtype=torch.int32, device="cuda")
    assert L.trag_merge_packed_device(C.c_void_p(gathered.data_ptr()), 2, B, k, C.c_void_p(out_r.data_ptr()),
                                      C.c_void_p(out_s.data_ptr()), C.c_void_p(out_n.data_ptr()), None) == 0
    torch.cuda.synchronize()
    for b in range(B):
        er, es = oracle.search(corpus, Q[b], k, 0)
        assert_same_ranking(out_r[b].cpu().numpy().astype(np.uint32), out_s[b].cpu().numpy(), er, es, f"merge {b}")
    # world size 1 through the host class: the exchange is the identity plus the row offset
    ex = ShardExchange(k, max_queries=B, device=torch.device("cuda"))
    rows, scores, cnt = stores[0].search_batch_rows(Q, k)
    o_r, o_s, o_n = torch.zeros_like(out_r), torch.zeros_like(out_s), torch.zeros_like(out_n)
    ex.exchange(torch.from_numpy(rows.astype(np.int32)).cuda(), torch.from_numpy(scores).cuda(),
                torch.from_numpy(cnt.astype(np.int32)).cuda(), 7, o_r, o_s, o_n)
    torch.cuda.synchronize()
    assert np.array_equal(o_r.cpu().numpy(), rows.astype(np.int32) + 7)
    assert o_s.cpu().numpy().tobytes() == scores.tobytes()


def test_config1_plumbing_shape(trag, oracle):
    """The reference's CPU-runnable case: a 10k-vector 384-d index, one query, top-10
    (MockEmbedder(384) rows are unit-norm pseudo-random vectors; their exact values
    are not pinned by the reference, so seeded unit rows stand in)."""
    n, d = 10_000, 384
    store = trag.VectorStore.with_dimension(d)
    corpus = oracle.synth_rows(1001, 0, n, d)
    chunks = [trag.Chunk(content=f"chunk {i}", embedding=corpus[i]) for i in range(n)]
    store.insert_batch(chunks)
    q = oracle.synth_rows(1101, 0, 1, d)[0]
    res = store.search(q, 10)
    er, es = oracle.search(corpus, q, 10, 0)
    assert [chunks.index(store.get(cid)) for cid, _ in res] == er.tolist()
    assert np.array([s for _, s in res], dtype=np.float32).tobytes() == es.tobytes()
    fr, fs = oracle.search_faithful(corpus, q, 10, 0)   # the reference-shaped loop agrees
    assert fr.tolist() == er.tolist() and fs.tobytes() == es.tobytes()


@pytest.mark.parametrize("metric", [0, 1])
def test_large_k_returns_every_row_sorted(trag, oracle, metric):
    """k far beyond the one-workgroup sort (and k > len): the reference never
    errors (src/index.rs:409), it returns min(k, len) rows in order."""
    n, d = 30_000, 128
    rng = np.random.default_rng(21)
    corpus = rng.standard_normal((n, d)).astype(np.float32)
    corpus[100] = corpus[20_000]
    store = make_store(trag, d, metric)
    store.insert_rows(corpus)
    store.remove_row(7)
    dead = np.zeros(n, dtype=np.uint8)
    dead[7] = 1
    q = corpus[100].copy()
    for k in (5000, 29_999, 50_000):
        rows, scores = store.search_rows(q, k)
        er, es = oracle.search(corpus, q, k, metric, dead=dead)
        assert len(rows) == min(k, n - 1)
        assert_same_ranking(rows, scores, er, es, f"large k={k} metric={metric}")
    rows, scores = store.search_rows(q, 10)          # the usual path still works afterwards
    er, es = oracle.search(corpus, q, 10, metric, dead=dead)
    assert_same_ranking(rows, scores, er, es, "after large k")


def test_config1_mock_embedder_corpus(trag, oracle):
    """The smallest benchmark configuration (10 000 x 384, top 10) with the reference's own generator: MockEmbedder(384) vectors of
    10k texts, one query, top-10 (benches/retrieval.rs shape, src/embed.rs:124-155 inputs)."""
    e = trag.MockEmbedder.new(384)
    n = 10_000
    corpus = np.stack(e.embed_batch([f"document {i}" for i in range(n)]))
    store = trag.VectorStore.with_dimension(384)
    store.insert_rows(corpus)
    for text in ("document 4242", "a question nobody indexed"):
        q = e.embed_query(text)
        rows, scores = store.search_rows(q, 10)
        er, es = oracle.search(corpus, q, 10, 0)
        assert rows.tolist() == er.tolist() and scores.tobytes() == es.tobytes()
    rows, scores = store.search_rows(e.embed_query("document 4242"), 1)
    assert rows[0] == 4242 and abs(scores[0] - 1.0) < 1e-5


# ---------------- multi-query exact scan (2..7 queries share corpus passes) ----------------

@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("n,d,B", [(5000, 384, 2), (5000, 384, 3), (20_000, 128, 4), (20_000, 100, 5),
                                   (7777, 768, 7), (300, 36, 6), (64, 4, 2)])
def test_multi_query_exact_scan(trag, oracle, metric, n, d, B):
    rng = np.random.default_rng(n + d + B + metric)
    corpus = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((B, d)).astype(np.float32)
    Q[0] = corpus[n // 2]                      # an exact hit
    store = make_store(trag, d, metric, EXACT)
    store.insert_rows(corpus)
    dead = np.zeros(n, dtype=np.uint8)
    for r in (0, n // 2 + 1, n - 1):
        store.remove_row(r)
        dead[r] = 1
    for k in (1, 10, 50):
        rows, scores, cnt = store.search_batch_rows(Q, k)
        for b in range(B):
            er, es = oracle.search(corpus, Q[b], k, metric, dead=dead)
            assert cnt[b] == len(er)
            assert_same_ranking(rows[b][:cnt[b]], scores[b][:cnt[b]], er, es, f"multi query {b} k={k}")
    assert store.last_stats()["fallbacks"] == 0


def test_multi_query_with_tie_groups_and_zero_query(trag, oracle):
    # 9000 identical rows force the multi-level select for the queries that hit them
    n, d = 12_000, 64
    rng = np.random.default_rng(1)
    corpus = rng.standard_normal((n, d)).astype(np.float32)
    corpus[1000:10_000] = corpus[1000]
    Q = rng.standard_normal((4, d)).astype(np.float32)
    Q[1] = corpus[1000]
    Q[2] = 0.0                                  # zero query: every cosine is 0.0
    store = make_store(trag, d, 0, EXACT)
    store.insert_rows(corpus)
    for _ in range(2):                          # the second call runs with the learned select plan
        rows, scores, cnt = store.search_batch_rows(Q, 10)
        for b in range(4):
            er, es = oracle.search(corpus, Q[b], 10, 0)
            assert_same_ranking(rows[b], scores[b], er, es, f"multi tie query {b}")


def test_multi_query_device_entry(trag, oracle):
    import torch
    n, d, B, k = 40_000, 384, 7, 10
    store = make_store(trag, d, 0, EXACT, reserve=n)
    store.fill_synthetic(n, 5)
    corpus = store.read_rows(0, n)
    Q = oracle.synth_rows(6, 0, B, d)
    dq = torch.from_numpy(Q).cuda()
    o_rows = torch.zeros((B, k), dtype=torch.int32, device="cuda")
    o_scores = torch.zeros((B, k), dtype=torch.float32, device="cuda")
    o_n = torch.zeros(B, dtype=torch.int32, device="cuda")
    o_st = torch.full((B,), 9, dtype=torch.int32, device="cuda")
    store.search_device(dq.data_ptr(), B, k, o_rows.data_ptr(), o_scores.data_ptr(), o_n.data_ptr(),
                        o_st.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert o_st.tolist() == [0] * B and o_n.tolist() == [k] * B
    for b in range(B):
        er, es = oracle.search(corpus, Q[b], k, 0)
        assert_same_ranking(o_rows[b].cpu().numpy().astype(np.uint32), o_scores[b].cpu().numpy(), er, es, f"dev {b}")


# ---------------- bf16 matrix-core prefilter + exact f32 finish ----------------

BF16 = 4


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("n,d,B", [(70_000, 384, 9), (100_000, 768, 40), (66_000, 64, 100), (80_000, 128, 200),
                                   (131_072, 256, 300), (70_001, 384, 130), (100_003, 384, 600),
                                   (66_000, 128, 2100)])
def test_bf16_prefilter_parity(trag, oracle, metric, n, d, B):
    rng = np.random.default_rng(n + d + B)
    corpus = unit_rows(rng, n, d) if metric == 0 else rng.standard_normal((n, d)).astype(np.float32)
    corpus[17] = corpus[n - 5]            # duplicate rows in different tiles
    corpus[1000] = 0.0                    # zero-norm row
    corpus[2000] *= 1e-3                  # a short row
    Q = rng.standard_normal((B, d)).astype(np.float32)
    Q[3] = corpus[17]                     # exact tie at the top
    Q[4] = corpus[4242] * np.float32(1.0001) + np.float32(1e-4)   # a near-duplicate
    Q[5] = 0.0                            # zero query: every cosine ties at 0.0 -> exact fallback
    store = make_store(trag, d, metric, BF16)
    store.insert_rows(corpus)
    dead = np.zeros(n, dtype=np.uint8)
    for r in (5, 64, 12_345, n - 5):
        store.remove_row(r)
        dead[r] = 1
    for k in (10, 100):
        rows, scores, cnt = store.search_batch_rows(Q, k)
        assert cnt.tolist() == [k] * B
        check = range(B) if B <= 40 else list(range(8)) + list(range(B - 8, B))
        for b in check:
            er, es = oracle.search(corpus, Q[b], k, metric, dead=dead)
            assert_same_ranking(rows[b], scores[b], er, es, f"bf16 n={n} d={d} q={b} k={k} metric={metric}")
        st = store.last_stats()
        assert st["mode_used"] == BF16
        assert st["fallbacks"] <= 3


def test_bf16_auto_and_limits(trag, oracle):
    n, d = 70_000, 384
    store = make_store(trag, d, 0, 0, reserve=n)
    store.fill_synthetic(n, 77)
    corpus = store.read_rows(0, n)
    Q = oracle.synth_rows(78, 0, 12, d)
    rows, scores, cnt = store.search_batch_rows(Q, 10)
    assert store.last_stats()["mode_used"] == BF16          # AUTO: more queries than one multi-query pass takes
    for b in range(12):
        er, es = oracle.search(corpus, Q[b], 10, 0)
        assert_same_ranking(rows[b], scores[b], er, es, f"auto bf16 {b}")
    # 2..8 queries: the multi-query exact pass or the prefilter, whichever is cheaper for the corpus' size --
    # on 70 k rows (a 30 us scan) the prefilter's 40 us of small launches cost more than seven extra queries
    r4, s4, _ = store.search_batch_rows(Q[:4], 10)
    assert store.last_stats()["mode_used"] == EXACT
    assert np.array_equal(r4, rows[:4]) and s4.tobytes() == scores[:4].tobytes()
    store.search_batch_rows(Q[:8], 10)
    assert store.last_stats()["mode_used"] == EXACT
    # ... on 1.5 M x 192 (a 190 us scan) four queries go to the prefilter, two stay with the multi-query pass
    big = make_store(trag, 192, 0, 0, reserve=1_500_000)
    big.fill_synthetic(1_500_000, 79)
    cb = big.read_rows(0, 1_500_000)
    Qb = oracle.synth_rows(80, 0, 4, 192)
    rb, sb, _ = big.search_batch_rows(Qb, 10)
    assert big.last_stats()["mode_used"] == BF16
    for b in range(4):
        er, es = oracle.search(cb, Qb[b], 10, 0)
        assert_same_ranking(rb[b], sb[b], er, es, f"auto bf16, four queries on a long corpus {b}")
    r2, s2, _ = big.search_batch_rows(Qb[:2], 10)
    assert big.last_stats()["mode_used"] == EXACT
    assert np.array_equal(r2, rb[:2]) and s2.tobytes() == sb[:2].tobytes()
    big.close()
    store.search_batch_rows(Q, 300)
    assert store.last_stats()["mode_used"] == MFMA          # k too large for the candidate lists
    # clustered scores: thousands of rows within 2 eps of the k-th best overflow the lists -> exact repeat
    near = (corpus[0][None, :] + 1e-3 * np.random.default_rng(1).standard_normal((6000, d))).astype(np.float32)
    store2 = make_store(trag, d, 0, BF16)
    c2 = np.concatenate([corpus[:66_000], near])
    store2.insert_rows(c2)
    Q2 = np.concatenate([corpus[0][None, :], Q[:9]]).astype(np.float32)
    rows, scores, cnt = store2.search_batch_rows(Q2, 10)
    for b in range(10):
        er, es = oracle.search(c2, Q2[b], 10, 0)
        assert_same_ranking(rows[b], scores[b], er, es, f"clustered bf16 {b}")
    assert store2.last_stats()["fallbacks"] >= 1


def test_bf16_clustered_corpus_switches_to_f32_matrix_cores(trag, oracle):
    """Every query sits in a cluster of thousands of near-identical rows: the bf16 lists overflow for
    the whole batch, the batch is repeated on the f32 matrix cores and AUTO remembers."""
    d, rng = 128, np.random.default_rng(21)
    centers = unit_rows(rng, 16, d)
    corpus = np.repeat(centers, 5000, axis=0) + 2e-3 * rng.standard_normal((80_000, d)).astype(np.float32)
    corpus = corpus.astype(np.float32)
    store = make_store(trag, d, 0, 0)
    store.insert_rows(corpus)
    Q = (centers + 1e-3 * rng.standard_normal((16, d))).astype(np.float32)
    rows, scores, _ = store.search_batch_rows(Q, 10)
    st = store.last_stats()
    assert st["mode_used"] == MFMA and st["fallbacks"] >= 2
    for b in range(16):
        er, es = oracle.search(corpus, Q[b], 10, 0)
        assert_same_ranking(rows[b], scores[b], er, es, f"clustered batch {b}")
    store.search_batch_rows(Q, 10)
    assert store.last_stats()["mode_used"] == MFMA          # learned: no second bf16 attempt


def test_concurrent_searches_on_one_handle(trag, oracle):
    """`search(&self)` is shared in the reference (src/index.rs:386): searches from several
    threads on one store must be safe (the handle serialises them) and return what a single
    thread gets; a second store searched at the same time must not disturb it."""
    import threading
    n, d = 30_000, 128
    rng = np.random.default_rng(314)
    corpus = unit_rows(rng, n, d)
    store = make_store(trag, d)
    store.insert_rows(corpus)
    other = make_store(trag, d, 2)
    other.insert_rows(corpus[:5000])
    Q = unit_rows(rng, 24, d)
    expect = [oracle.search(corpus, q, 10, 0) for q in Q]
    expect_other = [oracle.search(corpus[:5000], q, 5, 2) for q in Q]
    errors = []

    def worker(t):
        try:
            for rep in range(3):
                for b in range(t, 24, 4):
                    rows, scores = store.search_rows(Q[b], 10)
                    assert_same_ranking(rows, scores, expect[b][0], expect[b][1], f"thread {t} query {b}")
                    if t == 0:
                        r2, s2 = other.search_rows(Q[b], 5)
                        assert_same_ranking(r2, s2, expect_other[b][0], expect_other[b][1], f"other store {b}")
                    if t == 1:   # batches interleave with single queries on the same handle
                        rb, sb, _ = store.search_batch_rows(Q[b:b + 3], 10)
                        for j in range(rb.shape[0]):
                            assert_same_ranking(rb[j], sb[j], expect[b + j][0], expect[b + j][1], f"batch {b}+{j}")
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[0]


def device_searches_one_by_one(store, Q, k, sync=True):
    """Each query through trag_store_search_device on torch's current stream; statuses and results."""
    import torch
    nq = len(Q)
    dq = torch.from_numpy(Q).cuda()
    o_rows = torch.zeros((nq, k), dtype=torch.int32, device="cuda")
    o_scores = torch.zeros((nq, k), dtype=torch.float32, device="cuda")
    o_n = torch.zeros(nq, dtype=torch.int32, device="cuda")
    o_st = torch.full((nq,), 9, dtype=torch.int32, device="cuda")
    sp = torch.cuda.current_stream().cuda_stream
    for b in range(nq):
        store.search_device(dq[b].data_ptr(), 1, k, o_rows[b].data_ptr(), o_scores[b].data_ptr(),
                            o_n[b].data_ptr(), o_st[b].data_ptr(), sp)
        if sync:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    return o_st.tolist(), o_rows.cpu().numpy().astype(np.uint32), o_scores.cpu().numpy()


def test_device_entry_learns_the_multi_level_select(trag, oracle):
    """Callers of trag_store_search_device never hand a status back. On a corpus whose scores
    crowd one coarse bin (rows packed around one direction: every cosine lies in a narrow band) the
    first queries report status 3; the handle must notice by itself (pinned host word bumped by
    the select kernel) and serve later queries with the multi-level select."""
    n, d, k = 60_000, 384, 10
    rng = np.random.default_rng(2718)
    centre = unit_rows(rng, 1, d)[0]
    corpus = np.ascontiguousarray(centre[None, :] + 0.02 * rng.standard_normal((n, d)), dtype=np.float32)
    store = make_store(trag, d, 0)
    store.insert_rows(corpus)
    Q = np.ascontiguousarray(centre[None, :] + 0.02 * rng.standard_normal((24, d)), dtype=np.float32)
    st, rows, scores = device_searches_one_by_one(store, Q, k)
    assert st[0] == 3, st                 # crowded coarse bin
    assert st[-8:] == [0] * 8, st         # learned within a few queries
    for b in range(16, 24):
        er, es = oracle.search(corpus, Q[b], k, 0)
        assert_same_ranking(rows[b], scores[b], er, es, f"dev clustered {b}")


def test_device_entry_on_a_short_euclidean_store_needs_no_learning(trag, oracle):
    """Euclidean distances between unit vectors without near neighbours are all close to sqrt 2: one
    coarse bin holds nearly every row. A store too short for sparse key stores starts on the
    multi-level select, so even queries enqueued far ahead of any synchronisation finish (status 0)
    with the oracle's rows and score bits."""
    n, d, k = 100_000, 384, 10
    rng = np.random.default_rng(2719)
    corpus = unit_rows(rng, n, d)
    store = make_store(trag, d, 1)
    store.insert_rows(corpus)
    Q = unit_rows(rng, 24, d)
    st, rows, scores = device_searches_one_by_one(store, Q, k, sync=False)
    assert st == [0] * 24, st
    for b in (0, 1, 7, 23):
        er, es = oracle.search(corpus, Q[b], k, 1)
        assert_same_ranking(rows[b], scores[b], er, es, f"dev eu {b}")


def test_striped_store_on_one_gpu(trag, oracle):
    """ShardedVectorStore with the real device store as its shard (world size 1: the exchange
    degenerates to pack + merge on the GPU)."""
    import torch
    from trueno_rag_amd.sharded import ShardedVectorStore
    n, d = 3000, 64
    rng = np.random.default_rng(77)
    corpus = unit_rows(rng, n, d)
    st = ShardedVectorStore(make_store(trag, d), device=torch.device("cuda", torch.cuda.current_device()))
    assert st.insert_rows(corpus[:1000]) == 0 and st.insert_rows(corpus[1000:]) == 1000
    st.remove_row(42)
    dead = np.zeros(n, np.uint8)
    dead[42] = 1
    Q = unit_rows(rng, 5, d)
    rows, scores, counts = st.search_batch_rows(Q, 10)
    assert st.len() == n - 1
    for b in range(5):
        er, es = oracle.search(corpus, Q[b], 10, 0, dead=dead)
        assert counts[b] == 10
        assert_same_ranking(rows[b], scores[b], er, es, f"striped {b}")
