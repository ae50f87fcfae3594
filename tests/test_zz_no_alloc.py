"""trag_store_search_device_noalloc (the entry point that allocates nothing: its scratch is a fixed
device array inside the library) against trag_store_search_device, through the C ABI: rows, score
bits, counts and statuses must be identical, on two calls in a row, the second with a smaller
input (fewer rows, fewer queries, another store) so that whatever the first call left in the shared
array must not leak into it."""
import numpy as np
import pytest

from conftest import has_gpu
from util import assert_same_ranking

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs a GPU")]


def _both(trag, torch, store, Q, k):
    """(regular entry, no-alloc entry) results of the same queries on the same store."""
    dev = torch.device("cuda", store.config().device if store.config().device >= 0 else 0)
    B = Q.shape[0]
    dq = torch.from_numpy(Q).to(dev)
    outs = []
    for fn in (store.search_device, store.search_device_noalloc):
        rows = torch.full((B, k), 0x7FFFFFFF, dtype=torch.int32, device=dev)
        scores = torch.full((B, k), float("nan"), dtype=torch.float32, device=dev)
        n = torch.full((B,), -1, dtype=torch.int32, device=dev)
        status = torch.full((B,), 9, dtype=torch.int32, device=dev)
        stream = torch.cuda.current_stream(dev)
        fn(dq.data_ptr(), B, k, rows.data_ptr(), scores.data_ptr(), n.data_ptr(), status.data_ptr(),
           stream.cuda_stream)
        torch.cuda.synchronize(dev)
        outs.append((rows.cpu().numpy(), scores.cpu().numpy(), n.cpu().numpy(), status.cpu().numpy()))
    return outs


def _check(ref, got, what):
    assert np.array_equal(ref[2], got[2]), f"{what}: counts {got[2]} vs {ref[2]}"
    assert np.array_equal(ref[3], got[3]), f"{what}: statuses {got[3]} vs {ref[3]}"
    for b in range(len(ref[2])):
        if ref[3][b] != 0:
            continue
        c = int(ref[2][b])
        assert_same_ranking(got[0][b, :c], got[1][b, :c], ref[0][b, :c], ref[1][b, :c], f"{what}, query {b}")


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_noalloc_equals_regular_entry_two_calls_second_smaller(trag, oracle, metric):
    import torch
    d = 64
    big_n, small_n = 2_000_000, 30_000     # 2M x 64: 30 tiles per scan wave, the sparse key stores are on
    big = trag.VectorStore(trag.VectorStoreConfig(dimension=d, metric=trag.DistanceMetric(metric), device=0,
                                                  scan_mode=1, reserve_rows=big_n))
    big.fill_synthetic(big_n, 4242)
    small = trag.VectorStore(trag.VectorStoreConfig(dimension=d, metric=trag.DistanceMetric(metric), device=0,
                                                    scan_mode=1, reserve_rows=small_n))
    small.fill_synthetic(small_n, 777)
    for r in (5, 17, 29_999):
        small.remove_row(r)
    Q = oracle.synth_rows(9001, 0, 3, d)
    # first call: the large store, three queries, k = 10
    ref1, got1 = _both(trag, torch, big, Q, 10)
    _check(ref1, got1, f"metric {metric}, 2M rows")
    assert (ref1[3] == 0).all()
    # second call, smaller input: 30 k rows of another store, one query, k = 100 -- the array still holds
    # the first call's keys and tile maxima beyond row 30 000
    ref2, got2 = _both(trag, torch, small, Q[1:2], 100)
    _check(ref2, got2, f"metric {metric}, 30k rows after 2M")
    # ... and against the oracle, so that both entries are not wrong together
    corpus = small.read_rows(0, small_n)
    dead = np.zeros(small_n, dtype=np.uint8)
    dead[[5, 17, 29_999]] = 1
    er, es = oracle.search(corpus, Q[1], 100, metric, dead=dead)
    assert_same_ranking(got2[0][0, :len(er)], got2[1][0, :len(er)], er, es, "no-alloc entry vs oracle")
    # back to the large store: the small call's leftovers must not matter either
    ref3, got3 = _both(trag, torch, big, Q[2:3], 10)
    _check(ref3, got3, f"metric {metric}, 2M rows again")
    big.close()
    small.close()


def test_noalloc_limits_and_edge_cases(trag, oracle):
    import torch
    d = 32
    st = trag.VectorStore(trag.VectorStoreConfig(dimension=d, device=0, reserve_rows=6000))
    st.fill_synthetic(6000, 5)
    Q = oracle.synth_rows(6, 0, 2, d)
    # k above the one-workgroup sort's capacity needs the global sort's buffer: refused, not allocated
    dq = torch.from_numpy(Q).cuda()
    rows = torch.zeros((2, 5000), dtype=torch.int32, device="cuda")
    scores = torch.zeros((2, 5000), dtype=torch.float32, device="cuda")
    n = torch.zeros(2, dtype=torch.int32, device="cuda")
    with pytest.raises(trag.TragError) as e:
        st.search_device_noalloc(dq.data_ptr(), 2, 5000, rows.data_ptr(), scores.data_ptr(), n.data_ptr())
    assert e.value.status == 6 and "global sort" in str(e.value)
    # k = 0 and k > len behave as in the regular entry
    ref, got = _both(trag, torch, st, Q, 3)
    _check(ref, got, "k=3")
    for r in range(5990):
        st.remove_row(r)
    ref, got = _both(trag, torch, st, Q, 50)   # 10 live rows, k = 50
    _check(ref, got, "k > len")
    assert (got[2] == 10).all()
    # an empty store
    empty = trag.VectorStore(trag.VectorStoreConfig(dimension=d, device=0))
    ref, got = _both(trag, torch, empty, Q, 5)
    assert (got[2] == 0).all() and (ref[2] == 0).all()
    # a wrong dimension is the same error
    with pytest.raises(trag.DimensionMismatch):
        import ctypes as C
        L = trag.lib()
        from trueno_rag_amd import _abi as A
        A.check(L.trag_store_search_device_noalloc(st.handle, C.c_void_p(dq.data_ptr()), 1, d + 1, 3,
                                                   C.c_void_p(rows.data_ptr()), C.c_void_p(scores.data_ptr()),
                                                   C.c_void_p(n.data_ptr()), None, None), d, d + 1)
    st.close()
    empty.close()


def test_noalloc_first_call_on_a_short_euclidean_store(trag, oracle):
    """The no-allocation entry decides the select plan as the regular one does: a short Euclidean store
    (distances between normalised vectors crowd one coarse score bin) searched ONLY through it finishes
    every query, from the first one, with the oracle's rows and score bits."""
    import torch
    d, n = 64, 30_000
    st = trag.VectorStore(trag.VectorStoreConfig(dimension=d, metric=trag.DistanceMetric(1), device=0, reserve_rows=n))
    st.fill_synthetic(n, 99)
    Q = oracle.synth_rows(98, 0, 4, d)
    dq = torch.from_numpy(Q).cuda()
    rows = torch.zeros((4, 10), dtype=torch.int32, device="cuda")
    scores = torch.zeros((4, 10), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(4, dtype=torch.int32, device="cuda")
    status = torch.full((4,), 9, dtype=torch.int32, device="cuda")
    st.search_device_noalloc(dq.data_ptr(), 4, 10, rows.data_ptr(), scores.data_ptr(), cnt.data_ptr(),
                             status.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert status.tolist() == [0, 0, 0, 0] and cnt.tolist() == [10] * 4
    corpus = st.read_rows(0, n)
    for b in range(4):
        er, es = oracle.search(corpus, Q[b], 10, 1)
        assert_same_ranking(rows[b].cpu().numpy().astype(np.uint32), scores[b].cpu().numpy(), er, es, f"query {b}")
    st.close()


@pytest.mark.parametrize("metric", [0, 1])
def test_noalloc_captured_in_a_graph_and_replayed(trag, oracle, metric):
    """One call captured into a HIP graph on a store long enough for the sparse key stores (2M x 64:
    30 tiles per scan wave), then replayed with other queries in the same device buffer: every replay
    must return what the regular entry returns for those queries (a tag or a bound that survived from
    one replay to the next would make the select skip rows it needs)."""
    import torch
    d, n, k, B = 64, 2_000_000, 10, 2
    st = trag.VectorStore(trag.VectorStoreConfig(dimension=d, metric=trag.DistanceMetric(metric), device=0,
                                                 scan_mode=1, reserve_rows=n))
    st.fill_synthetic(n, 4243)
    Qs = oracle.synth_rows(9100, 0, 4 * B, d).reshape(4, B, d)
    dq = torch.from_numpy(Qs[0].copy()).cuda()
    rows = torch.zeros((B, k), dtype=torch.int32, device="cuda")
    scores = torch.zeros((B, k), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
    status = torch.full((B,), 9, dtype=torch.int32, device="cuda")
    # one regular search first: the store's derived data are up to date, nothing is pending
    st.search_rows(Qs[0][0], k)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        st.search_device_noalloc(dq.data_ptr(), B, k, rows.data_ptr(), scores.data_ptr(), cnt.data_ptr(),
                                 status.data_ptr(), torch.cuda.current_stream().cuda_stream)
    for rep in range(4):
        dq.copy_(torch.from_numpy(Qs[rep].copy()))
        rows.zero_(); scores.zero_(); cnt.zero_(); status.fill_(9)
        graph.replay()
        torch.cuda.synchronize()
        got = (rows.cpu().numpy().astype(np.uint32), scores.cpu().numpy(), cnt.cpu().numpy(), status.cpu().numpy())
        assert got[3].tolist() == [0] * B and got[2].tolist() == [k] * B, (rep, got[2], got[3])
        for b in range(B):
            er, es = st.search_rows(Qs[rep][b], k)      # the regular (host) entry, exact scan
            assert_same_ranking(got[0][b], got[1][b], er, es, f"replay {rep}, query {b}")
    # the sparse key stores were in force in the captured scans: the regular entry reports the bound
    assert st.last_stats()["keys_kept_from"] != 0
    st.close()
