"""GPU: batched Reciprocal Rank Fusion on the device (trag_fuse_rrf_device) against the oracle's
restatement of src/fusion.rs:68-84, and the hybrid step of src/retrieve.rs:175-220 with both
searches and the fusion enqueued on the device without a host round trip."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(t):
    import torch
    return torch.from_numpy(t).cuda()


@pytest.fixture
def rrf_device(trag):
    import importlib
    return importlib.import_module(trag.__name__ + ".fusion").rrf_device


def test_rrf_device_matches_oracle(trag, oracle, rrf_device):
    import torch
    rng = np.random.default_rng(12)
    B, sd, ss = 37, 50, 64
    dense = np.zeros((B, sd), dtype=np.uint32)
    sparse = np.zeros((B, ss), dtype=np.uint32)
    nd = rng.integers(0, sd + 1, B).astype(np.uint32)
    ns = rng.integers(0, ss + 1, B).astype(np.uint32)
    nd[0], ns[0] = 0, 0                      # both lists empty
    nd[1], ns[1] = sd, 0                     # one list empty
    nd[2], ns[2] = sd, ss
    for b in range(B):
        pool = rng.permutation(90 if b % 2 else 4000).astype(np.uint32)   # small pool: many shared ids
        dense[b, :nd[b]] = pool[:nd[b]]
        sparse[b, :ns[b]] = rng.permutation(pool)[:ns[b]]
    sparse[2, :5] = dense[2, :5]             # equal ranks in both lists: ties among fused scores
    for k_rrf in (60.0, 1.0):
        for k_out in (10, 200):
            out_ids = torch.full((B, k_out), 0x7FFFFFFF, dtype=torch.int32, device="cuda")
            out_scores = torch.full((B, k_out), float("nan"), dtype=torch.float32, device="cuda")
            out_n = torch.full((B,), -1, dtype=torch.int32, device="cuda")
            d_dense, d_sparse = _dev(dense.view(np.int32)), _dev(sparse.view(np.int32))
            d_nd, d_ns = _dev(nd.view(np.int32)), _dev(ns.view(np.int32))
            rrf_device(d_dense.data_ptr(), d_nd.data_ptr(), sd, d_sparse.data_ptr(), d_ns.data_ptr(), ss, B, k_rrf,
                       k_out, out_ids.data_ptr(), out_scores.data_ptr(), out_n.data_ptr(),
                       torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            ids, scores, cnt = out_ids.cpu().numpy(), out_scores.cpu().numpy(), out_n.cpu().numpy()
            for b in range(B):
                e_ids, e_scores = oracle.rrf(dense[b, :nd[b]].astype(np.uint64), sparse[b, :ns[b]].astype(np.uint64), k_rrf)
                m = min(k_out, len(e_ids))
                assert cnt[b] == m, (b, cnt[b], m)
                assert ids[b, :m].tolist() == [int(x) for x in e_ids[:m]], (b, k_rrf, k_out)
                assert scores[b, :m].tobytes() == e_scores[:m].tobytes()
    # full lists without count arrays; repeated ids inside one list accumulate like HashMap entries
    dense[3, 7] = dense[3, 2]
    d_dense = _dev(dense.view(np.int32))
    out_ids = torch.zeros((B, 30), dtype=torch.int32, device="cuda")
    out_scores = torch.zeros((B, 30), dtype=torch.float32, device="cuda")
    out_n = torch.zeros((B,), dtype=torch.int32, device="cuda")
    rrf_device(d_dense.data_ptr(), 0, sd, d_sparse.data_ptr(), 0, ss, B, 60.0, 30, out_ids.data_ptr(),
               out_scores.data_ptr(), out_n.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for b in (3, 5):
        e_ids, e_scores = oracle.rrf(dense[b].astype(np.uint64), sparse[b].astype(np.uint64), 60.0)
        assert out_ids[b].cpu().numpy().tolist() == [int(x) for x in e_ids[:30]]
        assert out_scores[b].cpu().numpy().tobytes() == e_scores[:30].tobytes()
    with pytest.raises(trag.TragError):
        rrf_device(d_dense.data_ptr(), 0, 2000, d_sparse.data_ptr(), 0, ss, B, 60.0, 30, out_ids.data_ptr(),
                   out_scores.data_ptr(), out_n.data_ptr(), 0)


def test_hybrid_step_stays_on_the_device(trag, oracle, rrf_device):
    """Dense top-50 (one batched device-entry search), BM25 top-50 per query (device entry) and the
    fusion, all enqueued on one stream; one synchronisation at the end. Compared with the host route:
    oracle dense search + oracle BM25 + oracle RRF."""
    import torch
    n, d, B, cand, k = 20_000, 64, 6, 50, 10
    vocab, tokens = 3000, 24
    store = trag.VectorStore(trag.VectorStoreConfig(dimension=d, reserve_rows=n))
    store.fill_synthetic(n, 31)
    corpus = store.read_rows(0, n)
    bm25 = trag.BM25Index.new()
    bm25.fill_synthetic(n, vocab, tokens, 32)
    toks = oracle.bm25_synth_docs(32, 0, n, vocab, tokens)
    off = np.arange(n + 1, dtype=np.uint64) * tokens
    Q = oracle.synth_rows(33, 0, B, d)
    rng = np.random.default_rng(34)
    q_terms = [rng.integers(0, 200, 4).astype(np.uint32) for _ in range(B)]
    bm25.search_ids(q_terms[0], cand)        # builds the inverted lists
    dq = torch.from_numpy(Q).cuda()
    d_rows = torch.zeros((B, cand), dtype=torch.int32, device="cuda")
    d_scores = torch.zeros((B, cand), dtype=torch.float32, device="cuda")
    d_n = torch.zeros((B,), dtype=torch.int32, device="cuda")
    d_st = torch.zeros((B,), dtype=torch.int32, device="cuda")
    s_docs = torch.zeros((B, cand), dtype=torch.int32, device="cuda")
    s_scores = torch.zeros((B, cand), dtype=torch.float32, device="cuda")
    s_n = torch.zeros((B,), dtype=torch.int32, device="cuda")
    s_st = torch.zeros((B,), dtype=torch.int32, device="cuda")
    f_ids = torch.zeros((B, k), dtype=torch.int32, device="cuda")
    f_scores = torch.zeros((B, k), dtype=torch.float32, device="cuda")
    f_n = torch.zeros((B,), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    store.search_device(dq.data_ptr(), B, cand, d_rows.data_ptr(), d_scores.data_ptr(), d_n.data_ptr(),
                        d_st.data_ptr(), st)
    for b in range(B):
        bm25.search_ids_device(q_terms[b], cand, s_docs[b].data_ptr(), s_scores[b].data_ptr(),
                               s_n[b:].data_ptr(), s_st[b:].data_ptr(), st)
    rrf_device(d_rows.data_ptr(), d_n.data_ptr(), cand, s_docs.data_ptr(), s_n.data_ptr(), cand, B, 60.0, k,
               f_ids.data_ptr(), f_scores.data_ptr(), f_n.data_ptr(), st)
    torch.cuda.synchronize()
    assert d_st.cpu().tolist() == [0] * B
    for b in range(B):
        assert s_st[b].item() in (0, 3)
        if s_st[b].item() == 3:               # crowded score bins: the host entry finishes such a query
            docs, _ = bm25.search_ids(q_terms[b], cand)
            o_docs, _ = oracle.bm25_search(toks.ravel(), off, q_terms[b], cand)
            assert docs.tolist() == o_docs.tolist()
            continue
        er, es = oracle.search(corpus, Q[b], cand, 0)
        o_docs, o_scores = oracle.bm25_search(toks.ravel(), off, q_terms[b], cand)
        assert s_docs[b, :s_n[b].item()].cpu().numpy().astype(np.uint32).tolist() == o_docs.tolist()
        e_ids, e_scores = oracle.rrf(er.astype(np.uint64), o_docs.astype(np.uint64), 60.0)
        m = min(k, len(e_ids))
        assert f_n[b].item() == m
        assert f_ids[b, :m].cpu().numpy().tolist() == [int(x) for x in e_ids[:m]]
        assert f_scores[b, :m].cpu().numpy().tobytes() == e_scores[:m].tobytes()
    bm25.close()
