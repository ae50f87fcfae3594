"""GPU: the BM25 sparse index (trag_bm25_* through the C ABI) against the
reference's unit tests and, bit for bit, against the CPU oracle."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "ref_bm25_vectors.json")))


def docs_of(case):
    if "docs_rule" in case:
        return [f"document {i} about rust" for i in range(10)]
    return case["docs"]


def build(trag, docs, **kw):
    ix = trag.BM25Index(**kw)
    chunks = [trag.Chunk(content=d) for d in docs]
    for c in chunks:
        ix.add(c)
    return ix, chunks


def test_defaults(trag):
    ix = trag.BM25Index.new()
    assert ix.len() == 0 and ix.is_empty()
    assert abs(ix.k1 - 1.2) < 0.01 and abs(ix.b - 0.75) < 0.01
    ix2 = trag.BM25Index.with_params(1.5, 0.5)
    assert abs(ix2.k1 - 1.5) < 0.01 and abs(ix2.b - 0.5) < 0.01


@pytest.mark.parametrize("case", GOLD["tokenize"], ids=lambda c: c["name"])
def test_tokenize(trag, case):
    toks = trag.BM25Index.new().tokenize(case["text"])
    for t in case["contains"]:
        assert t in toks
    for t in case["not_contains"]:
        assert t not in toks


@pytest.mark.parametrize("case", GOLD["add"], ids=lambda c: c["name"])
def test_add(trag, case):
    ix = trag.BM25Index.new()
    ix.add_batch([trag.Chunk(content=d) for d in case["docs"]])
    assert ix.len() == case["expect_len"] and not ix.is_empty()
    for t in case["indexed_terms"]:
        assert t in ix._terms


@pytest.mark.parametrize("case", GOLD["search"], ids=lambda c: c["name"])
def test_search_cases(trag, case):
    from oracle import bm25_text
    ix, chunks = build(trag, docs_of(case))
    res = ix.search(case["query"], case["k"])
    if "expect_len" in case:
        assert len(res) == case["expect_len"]
    if case.get("expect_nonempty"):
        assert res
    ids = [r[0] for r in res]
    for d in case.get("expect_contains", []):
        assert chunks[d].id in ids
    if "expect_first" in case:
        assert ids[0] == chunks[case["expect_first"]].id
    # and exactly the oracle's ranking and scores
    ox = bm25_text.TextIndex()
    for d in docs_of(case):
        ox.add(d)
    od, os_ = ox.search(case["query"], case["k"])
    assert [chunks[int(d)].id for d in od] == ids
    assert np.asarray([r[1] for r in res], dtype=np.float32).tobytes() == os_.tobytes()


def test_remove_case(trag):
    case = GOLD["remove"][0]
    ix, chunks = build(trag, case["docs"])
    assert ix.len() == 1
    ix.remove(chunks[0].id)
    assert ix.len() == case["expect_len_after"]
    assert ix.search(case["query"], case["k"]) == []


def same(got, want):
    gd, gs = got
    wd, ws = want
    assert gd.tolist() == wd.tolist()
    assert gs.tobytes() == ws.tobytes()


@pytest.mark.parametrize("n_docs,vocab,length", [(1, 5, 4), (63, 20, 8), (1000, 200, 16), (20000, 3000, 32),
                                                  (150000, 20000, 24)])
def test_parity_synthetic(trag, oracle, n_docs, vocab, length):
    toks = oracle.bm25_synth_docs(77, 0, n_docs, vocab, length)
    off = np.arange(n_docs + 1, dtype=np.uint64) * length
    ix = trag.BM25Index.new()
    ix.add_token_ids(toks.ravel(), off)
    assert ix.len() == n_docs
    rng = np.random.default_rng(n_docs)
    queries = [[0], [vocab - 1], [1, 1, 1], [0, 1, 2, 3], [vocab + 5], [vocab // 2, 0, vocab // 2, 7 % vocab],
               rng.integers(0, vocab, 8).tolist(), rng.integers(0, min(vocab, 50), 12).tolist()]
    for q in queries:
        for k in (1, 10, 50, 100):
            same(ix.search_ids(q, k), oracle.bm25_search(toks.ravel(), off, q, k))
    ix.close()


def test_parity_params_and_ragged(trag, oracle):
    rng = np.random.default_rng(9)
    n_docs = 5000
    lens = rng.integers(0, 40, n_docs)          # some documents are empty
    off = np.zeros(n_docs + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    terms = rng.zipf(1.3, int(off[-1])).astype(np.uint32) % 500
    for k1, b in ((1.2, 0.75), (1.5, 0.5), (0.0, 0.0), (2.0, 1.0)):
        ix = trag.BM25Index(k1, b)
        ix.add_token_ids(terms, off)
        for q in ([1], [2, 3, 4], [400, 1, 499], [7, 7]):
            same(ix.search_ids(q, 25), oracle.bm25_search(terms, off, q, 25, k1, b))
        ix.close()


def test_remove_and_add_between_searches(trag, oracle):
    toks = oracle.bm25_synth_docs(5, 0, 3000, 300, 20)
    off = np.arange(3001, dtype=np.uint64) * 20
    ix = trag.BM25Index.new()
    ix.add_token_ids(toks[:2000].ravel(), off[:2001])
    q = [0, 9, 120]
    same(ix.search_ids(q, 30), oracle.bm25_search(toks[:2000].ravel(), off[:2001], q, 30))
    removed = np.zeros(3000, dtype=np.uint8)
    for d in (0, 5, 1999, 700, 701, 5):
        ix.remove_doc(d)
        removed[d] = 1
    assert ix.len() == 2000 - 5
    same(ix.search_ids(q, 30), oracle.bm25_search(toks[:2000].ravel(), off[:2001], q, 30, removed=removed[:2000]))
    ix.add_token_ids(toks[2000:].ravel(), off[:1001])
    assert ix.len() == 3000 - 5
    same(ix.search_ids(q, 30), oracle.bm25_search(toks.ravel(), off, q, 30, removed=removed))
    ix.remove_doc(99999)  # unknown document: not an error
    ix.close()


def test_huge_tie_group(trag, oracle):
    # 6000 identical documents: every score ties; the lowest document indices win
    n = 6000
    terms = np.tile(np.array([3, 4, 4], dtype=np.uint32), n)
    off = np.arange(n + 1, dtype=np.uint64) * 3
    ix = trag.BM25Index.new()
    ix.add_token_ids(terms, off)
    d, s = ix.search_ids([4], 10)
    assert d.tolist() == list(range(10))
    same((d, s), oracle.bm25_search(terms, off, [4], 10))
    # and a second query on the same handle (the select it learned keeps working)
    same(ix.search_ids([3, 4], 7), oracle.bm25_search(terms, off, [3, 4], 7))
    ix.close()


def test_k_larger_than_matches_and_zero(trag, oracle):
    toks = oracle.bm25_synth_docs(2, 0, 500, 400, 6)
    off = np.arange(501, dtype=np.uint64) * 6
    ix = trag.BM25Index.new()
    ix.add_token_ids(toks.ravel(), off)
    d, s = ix.search_ids([399], 100)       # a rare term: fewer than k matches
    same((d, s), oracle.bm25_search(toks.ravel(), off, [399], 100))
    assert len(ix.search_ids([0], 0)[0]) == 0
    assert len(ix.search_ids([], 10)[0]) == 0
    ix.close()


def test_errors(trag):
    with pytest.raises(trag.InvalidConfig):
        trag.BM25Index(1.2, 1.5)
    ix = trag.BM25Index.new()
    with pytest.raises(trag.TragError):
        ix.add_token_ids(np.array([1 << 29], dtype=np.uint32), np.array([0, 1], dtype=np.uint64))
    assert ix.len() == 0
    assert ix.search("anything", 5) == []
    ix.close()


def test_fill_synthetic_matches_oracle_generator(trag, oracle):
    ix = trag.BM25Index.new()
    ix.fill_synthetic(70000, 5000, 16, 1005, first_global_doc=10)
    toks = oracle.bm25_synth_docs(1005, 10, 70000, 5000, 16)
    off = np.arange(70001, dtype=np.uint64) * 16
    for q in ([0, 1], [4999, 17, 300], [2500]):
        same(ix.search_ids(q, 50), oracle.bm25_search(toks.ravel(), off, q, 50))
    st = ix.stats()
    assert st["terms"] == len(np.unique(toks))
    ix.close()


def test_hybrid_retriever_with_bm25(trag, oracle):
    """src/retrieve.rs:175-220 with both halves on the GPU."""
    rng = np.random.default_rng(4)
    words = ["alpha", "beta", "gamma", "delta", "epsilon", "zeta", "kappa", "lambda", "sigma", "omega"]
    n, d = 300, 32
    emb = rng.standard_normal((n, d)).astype(np.float32)
    texts = [" ".join(rng.choice(words, rng.integers(3, 9))) for _ in range(n)]

    class Embedder:
        def embed_query(self, q):
            return emb[7] * 0.5 + emb[11] * 0.25

        def dimension(self):
            return d

    store = trag.VectorStore.with_dimension(d)
    r = trag.HybridRetriever.new(store, trag.BM25Index.new(), Embedder())
    chunks = [trag.Chunk(content=t).with_embedding(e) for t, e in zip(texts, emb)]
    r.index_batch(chunks)
    out = r.retrieve("alpha omega omega", 10)
    assert len(out) == 10
    # the same through the oracles
    from oracle import bm25_text
    ox = bm25_text.TextIndex()
    for t in texts:
        ox.add(t)
    sd, ss = ox.search("alpha omega omega", 50)
    dd, ds = oracle.search(emb, Embedder().embed_query(""), 50, 0)
    ids, scores = oracle.rrf(dd.astype(np.uint64), sd.astype(np.uint64), 60.0)
    assert [chunks[int(i)].id for i in ids[:10]] == [x.chunk.id for x in out]
    assert np.asarray([x.fused_score for x in out], dtype=np.float32).tobytes() == scores[:10].tobytes()
    sparse = {chunks[int(i)].id: float(s) for i, s in zip(sd, ss)}
    for x in out:
        assert x.sparse_score == sparse.get(x.chunk.id)


def test_reference_bench_shape(trag):
    """benches/retrieval.rs:44-68: 1000 documents "Document {i} about topic {i % 100} with
    keywords", query "topic keywords", k = 10 and 100; every document matches."""
    from oracle import bm25_text
    docs = [f"Document {i} about topic {i % 100} with keywords" for i in range(1000)]
    ix, chunks = build(trag, docs)
    ox = bm25_text.TextIndex()
    for d in docs:
        ox.add(d)
    for k in (10, 100):
        res = ix.search("topic keywords", k)
        od, os_ = ox.search("topic keywords", k)
        assert len(res) == k
        assert [chunks[int(d)].id for d in od] == [r[0] for r in res]
        assert np.asarray([r[1] for r in res], dtype=np.float32).tobytes() == os_.tobytes()


def test_sparse_retriever_reference_unit_tests(trag):
    """SparseRetriever, src/retrieve.rs:300-335, with the reference's own unit tests (:568-658) over the
    GPU-resident BM25 index."""
    r = trag.SparseRetriever.new()
    assert r.retrieve("test", 10) == []                                   # test_sparse_retriever_new / _default
    r.index(trag.Chunk("machine learning test"))
    res = r.retrieve("machine", 10)
    assert len(res) == 1 and res[0].sparse_score is not None              # test_sparse_retriever_index
    assert res[0].dense_score is None and res[0].best_score() == res[0].sparse_score
    m = trag.SparseRetriever.with_index(trag.BM25Index.with_params(1.5, 0.5))
    a, b = trag.Chunk("rust programming language"), trag.Chunk("python programming language")
    m.index(a)
    m.index(b)
    res = m.retrieve("programming", 10)
    assert len(res) == 2 and {x.chunk.id for x in res} == {a.id, b.id}    # test_sparse_retriever_multiple
    assert m.retrieve("programming", 1)[0].chunk.id == a.id               # equal scores: insertion order
