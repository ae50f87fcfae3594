"""Contract of the C ABI around the search itself: calls on different streams against one handle,
any k, ingestion one chunk at a time (the shape of HybridRetriever::index, src/retrieve.rs:156-164)."""
import time

import numpy as np
import pytest

from conftest import has_gpu
from util import assert_same_ranking

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs a GPU")]


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_device_entry_on_a_user_stream_then_host_entry(trag, oracle, mode):
    """A handle has one scratch. A device-entry search enqueued on the caller's stream and a
    host-entry search issued right behind it (the handle's own stream) must not overlap on the
    GPU: both results equal the oracle's."""
    import torch
    n, d, k = 3_000_000, 64, 10      # a scan long enough (~0.15 ms) to still be running when the second call starts
    store = trag.VectorStore(trag.VectorStoreConfig(dimension=d, device=0, scan_mode=mode, reserve_rows=n))
    store.fill_synthetic(n, 71)
    corpus = store.read_rows(0, n)
    Q = oracle.synth_rows(72, 0, 8, d)
    dq = torch.from_numpy(Q).cuda()
    user = torch.cuda.Stream()
    other = torch.cuda.Stream()
    torch.cuda.synchronize()
    for rep in range(4):
        a, b, c = Q[2 * rep % 8], Q[(2 * rep + 1) % 8], Q[(2 * rep + 3) % 8]
        rows = torch.zeros((2, k), dtype=torch.int32, device="cuda")
        scores = torch.zeros((2, k), dtype=torch.float32, device="cuda")
        cnt = torch.zeros(2, dtype=torch.int32, device="cuda")
        status = torch.full((2,), 9, dtype=torch.int32, device="cuda")
        ia, ic = 2 * rep % 8, (2 * rep + 3) % 8
        store.search_device(dq.data_ptr() + ia * d * 4, 1, k, rows.data_ptr(), scores.data_ptr(), cnt.data_ptr(),
                            status.data_ptr(), user.cuda_stream)
        hr, hs = store.search_rows(b, k)                      # host entry, at once
        store.search_device(dq.data_ptr() + ic * d * 4, 1, k, rows.data_ptr() + k * 4, scores.data_ptr() + k * 4,
                            cnt.data_ptr() + 4, status.data_ptr() + 4, other.cuda_stream)   # a second user stream
        torch.cuda.synchronize()
        assert status.cpu().tolist() == [0, 0]
        for got_r, got_s, q in ((rows[0].cpu().numpy(), scores[0].cpu().numpy(), a), (hr, hs, b),
                                (rows[1].cpu().numpy(), scores[1].cpu().numpy(), c)):
            er, es = oracle.search(corpus, q, k, 0)
            assert_same_ranking(got_r.astype(np.uint32), got_s, er, es, f"mode {mode} rep {rep}")
    store.close()


def test_bm25_device_entry_then_host_entry(trag):
    """The same for the sparse index (one accumulator array per handle)."""
    import torch
    bm = trag.BM25Index(device=0)
    bm.fill_synthetic(400_000, 5000, 32, 9)
    ref = [bm.search_ids([3, 40, 7, 900], 10), bm.search_ids([1, 2, 3, 4, 5], 10)]
    user = torch.cuda.Stream()
    import ctypes as C
    from trueno_rag_amd import _abi
    for rep in range(4):
        docs = torch.zeros(10, dtype=torch.int32, device="cuda")
        scores = torch.zeros(10, dtype=torch.float32, device="cuda")
        cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
        status = torch.full((1,), 9, dtype=torch.int32, device="cuda")
        q = np.array([3, 40, 7, 900], dtype=np.uint32)
        _abi.check(trag.lib().trag_bm25_search_device(bm.handle, q.ctypes.data_as(C.POINTER(C.c_uint32)), 4, 10,
                                                      C.c_void_p(docs.data_ptr()), C.c_void_p(scores.data_ptr()),
                                                      C.c_void_p(cnt.data_ptr()), C.c_void_p(status.data_ptr()),
                                                      C.c_void_p(user.cuda_stream)))
        h_docs, h_scores = bm.search_ids([1, 2, 3, 4, 5], 10)
        torch.cuda.synchronize()
        if int(status.item()) == 0:     # (3 = this query wants the multi-level select: the host entry's business)
            assert docs.cpu().numpy().astype(np.uint32).tolist() == ref[0][0].tolist()
            assert scores.cpu().numpy().tobytes() == ref[0][1].tobytes()
        assert h_docs.tolist() == ref[1][0].tolist() and h_scores.tobytes() == ref[1][1].tobytes()
    bm.close()


def test_any_k_is_legal(trag, oracle):
    """The reference truncates to k (src/index.rs:409): 'give me everything' is a huge k."""
    import ctypes as C
    d = 8
    store = trag.VectorStore(trag.VectorStoreConfig(dimension=d, device=0))
    rows = oracle.synth_rows(3, 0, 3, d)
    store.insert_rows(rows)
    q = oracle.synth_rows(4, 0, 1, d)[0]
    r, s = store.search_rows(q, 2 ** 31)
    er, es = oracle.search(rows, q, 3, 0)
    assert_same_ranking(r, s, er, es, "k = 2^31, three rows")
    # straight through the C ABI with arrays of 3 entries and a stride the library must not use up
    out_r = (C.c_uint32 * 3)()
    out_s = (C.c_float * 3)()
    n = C.c_uint32()
    qq = np.ascontiguousarray(q)
    rc = trag.lib().trag_store_search(store.handle, qq.ctypes.data_as(C.POINTER(C.c_float)), d, 0xFFFFFFFF, out_r,
                                      out_s, C.byref(n))
    assert rc == 0 and n.value == 3 and list(out_r) == er.tolist()
    bm = trag.BM25Index(device=0)
    bm.add_token_ids(np.array([1, 2, 3, 1, 2, 1], dtype=np.uint32), np.array([0, 3, 5, 6], dtype=np.uint64))
    docs, sc = bm.search_ids([1], 2 ** 31)
    assert sorted(docs.tolist()) == [0, 1, 2]
    store.close()
    bm.close()


def test_chunk_by_chunk_ingest_does_not_wait_for_the_gpu(trag, oracle):
    """100 000 single-row inserts (the retrievers' index() loop): staged through pinned memory,
    no synchronisation per call; the rows' norms are brought up to date before the next search."""
    import ctypes as C
    n, d, k = 100_000, 64, 10
    rows = oracle.synth_rows(55, 0, n, d)
    store = trag.VectorStore(trag.VectorStoreConfig(dimension=d, device=0))
    L, h = trag.lib(), store.handle
    fp = C.POINTER(C.c_float)
    first = C.c_uint32()
    base = rows.ctypes.data
    t0 = time.perf_counter()
    for i in range(n):
        rc = L.trag_store_insert(h, C.cast(base + i * d * 4, fp), 1, d, C.byref(first))
        if rc != 0 or first.value != i:
            raise AssertionError((i, rc, first.value))
    elapsed = time.perf_counter() - t0
    assert store.len() == n
    Q = oracle.synth_rows(56, 0, 20, d)
    for mode in (0, 1, 2):
        store.set_scan_mode(mode)
        for b in range(3):
            r, s = store.search_rows(Q[b], k)
            er, es = oracle.search(rows, Q[b], k, 0)
            assert_same_ranking(r, s, er, es, f"after single-row ingest, mode {mode}")
    store.set_scan_mode(0)
    rb, sb, nb = store.search_batch_rows(Q, k)       # the matrix-core route reads the cached norms
    for b in range(20):
        er, es = oracle.search(rows, Q[b], k, 0)
        assert_same_ranking(rb[b], sb[b], er, es, f"batch after single-row ingest, query {b}")
    assert np.array_equal(store.read_rows(n - 5, 5), rows[-5:])
    # interleaved with searches, a mirror and a removal
    store.set_bf16_mirror(True)
    extra = oracle.synth_rows(57, 0, 300, d)
    for i in range(300):
        store.insert_rows(extra[i:i + 1])
        if i % 100 == 50:
            store.remove_row(n + i - 1)
    allrows = np.concatenate([rows, extra])
    dead = np.zeros(n + 300, np.uint8)
    dead[[n + 49, n + 149, n + 249]] = 1
    for b in range(3):
        r, s = store.search_rows(extra[b * 7], k)
        er, es = oracle.search(allrows, extra[b * 7], k, 0, dead=dead)
        assert_same_ranking(r, s, er, es, "mirror scan after interleaved ingest")
    store.close()
    assert elapsed < 2.0, f"100k single-row inserts took {elapsed:.2f} s"


def test_clone_is_an_independent_copy(trag, oracle):
    """`#[derive(Clone)]` of the reference's store (src/index.rs:322): a clone answers like its
    source (rows, removed rows, mirror setting, deferred single-row inserts included) and the two
    diverge freely afterwards."""
    rng = np.random.default_rng(5)
    n, d = 5000, 64
    corpus = rng.standard_normal((n, d)).astype(np.float32)
    store = trag.VectorStore.with_dimension(d)
    store.insert_rows(corpus[:-3])
    for r in corpus[-3:]:
        store.insert_rows(r[None, :])            # small inserts: finished by the next search -- or by clone
    store.set_bf16_mirror(True)
    dead = np.zeros(n, dtype=np.uint8)
    for r in (0, 77, n - 1):
        store.remove_row(r)
        dead[r] = 1
    twin = store.clone()
    assert "len=%d" % (n - 3) in repr(twin) and twin.len() == n - 3
    assert twin.read_rows(0, n).tobytes() == corpus.tobytes()
    q = rng.standard_normal(d).astype(np.float32)
    er, es = oracle.search(corpus, q, 10, 0, dead=dead)
    for s in (store, twin):
        rows, scores = s.search_rows(q, 10)
        assert np.array_equal(rows, er) and scores.tobytes() == es.tobytes()
    twin.remove_row(int(er[0]))
    twin.insert_rows(q[None, :])
    rows, scores = twin.search_rows(q, 2)
    assert rows[0] == n and rows[1] == er[1]
    rows, scores = store.search_rows(q, 10)      # the source never saw any of that
    assert np.array_equal(rows, er) and store.len() == n - 3 and twin.len() == n - 3


@pytest.mark.parametrize("mirror", [False, True])
@pytest.mark.parametrize("pipe_mode", [1, 2])
@pytest.mark.parametrize("metric", [0, 1])
def test_pipelined_searches_equal_the_oracle_after_the_fence(trag, oracle, metric, pipe_mode, mirror):
    """Pipelined modes (trag_store_set_pipelined): the select of query i runs beside the scan of query
    i + 1 -- on the handle's side stream (1), or as one more workgroup of that scan's launch (2). 40
    single-query searches enqueued back to back on a user stream, a host-entry search and a batch in
    between (other routes: they wait for the pipeline), a second user stream; outputs are read only after
    trag_store_fence -- all equal the oracle's. (mirror: the store holds a bf16 copy, single queries take
    the scan of the copy, whose prune / exact scores / ordering go to the side stream in either mode; in the
    middle of the run the copy is dropped, so exact-route queries follow mirror-route ones on the same sets.)"""
    import torch
    n, d, k, B = 1_500_000, 64, 10, 40
    store = trag.VectorStore(trag.VectorStoreConfig(dimension=d, metric=trag.DistanceMetric(metric), device=0,
                                                    reserve_rows=n))
    store.fill_synthetic(n, 171)
    if mirror:
        store.set_bf16_mirror(True)
    corpus = store.read_rows(0, n)
    Q = oracle.synth_rows(172, 0, B, d)
    dq = torch.from_numpy(Q).cuda()
    rows = torch.zeros((B, k), dtype=torch.int32, device="cuda")
    scores = torch.zeros((B, k), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
    status = torch.full((B,), 9, dtype=torch.int32, device="cuda")
    user, other = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    store.set_pipelined(pipe_mode)
    host = {}
    for b in range(B):
        st = other if b in (17, 18) else user          # a change of stream in the middle of the run
        store.search_device(dq.data_ptr() + b * d * 4, 1, k, rows.data_ptr() + b * k * 4, scores.data_ptr() + b * k * 4,
                            cnt.data_ptr() + b * 4, status.data_ptr() + b * 4, st.cuda_stream)
        if b == 9:
            host[9] = store.search_rows(Q[9], k)        # host entry: not pipelined, waits for the pipeline
        if b == 25:
            host["batch"] = store.search_batch_rows(Q[:12], k)
        if b == 30 and mirror:
            store.set_bf16_mirror(False)
    # results may be read on a stream only behind the fence
    store.fence(user.cuda_stream)
    out = torch.empty_like(rows)
    with torch.cuda.stream(user):
        out.copy_(rows)
    user.synchronize()
    other.synchronize()
    torch.cuda.synchronize()
    assert status.cpu().tolist() == [0] * B and cnt.cpu().tolist() == [k] * B
    assert torch.equal(out.cpu()[:17], rows.cpu()[:17]) and torch.equal(out.cpu()[19:], rows.cpu()[19:])
    want = [oracle.search(corpus, Q[b], k, metric) for b in range(B)]
    for b in range(B):
        assert_same_ranking(rows[b].cpu().numpy().astype(np.uint32), scores[b].cpu().numpy(), want[b][0], want[b][1],
                            f"pipelined query {b}")
    assert_same_ranking(host[9][0], host[9][1], want[9][0], want[9][1], "host entry inside the pipeline")
    for b in range(12):
        assert_same_ranking(host["batch"][0][b], host["batch"][1][b], want[b][0], want[b][1], f"batch query {b}")
    # switched off again: single-stream behaviour, no fence needed
    store.set_pipelined(False)
    rows.zero_()
    store.search_device(dq.data_ptr(), 1, k, rows.data_ptr(), scores.data_ptr(), cnt.data_ptr(), status.data_ptr(),
                        user.cuda_stream)
    user.synchronize()
    assert_same_ranking(rows[0].cpu().numpy().astype(np.uint32), scores[0].cpu().numpy(), want[0][0], want[0][1], "off")
    store.close()


@pytest.mark.parametrize("pipe_mode", [1, 2])
def test_pipelined_outputs_are_not_ordered_on_the_callers_stream_without_the_fence(trag, oracle, pipe_mode):
    """What the fence is for: in pipelined mode the select writes the outputs on the handle's side stream,
    so a copy enqueued on the caller's stream right behind the call -- no fence -- is not ordered behind
    it and can read the outputs before they are written (it runs as soon as the scan has finished, beside
    the select). The same copy behind trag_store_fence always sees them. (Mode 2: the select is not even
    enqueued before the fence or the next query asks for it.)"""
    import torch
    n, d, k = 3_000_000, 64, 10
    store = trag.VectorStore(trag.VectorStoreConfig(dimension=d, device=0, reserve_rows=n))
    store.fill_synthetic(n, 173)
    q = oracle.synth_rows(174, 0, 1, d)
    dq = torch.from_numpy(q).cuda()
    want_rows, _ = store.search_rows(q[0], k)
    store.set_pipelined(pipe_mode)
    user = torch.cuda.Stream()
    stale_seen, fenced_ok = 0, 0
    for rep in range(8):
        rows = torch.full((k,), -1, dtype=torch.int32, device="cuda")
        scores = torch.zeros(k, dtype=torch.float32, device="cuda")
        cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
        early, late = torch.zeros_like(rows), torch.zeros_like(rows)
        torch.cuda.synchronize()
        store.search_device(dq.data_ptr(), 1, k, rows.data_ptr(), scores.data_ptr(), cnt.data_ptr(), 0, user.cuda_stream)
        with torch.cuda.stream(user):
            early.copy_(rows)                  # no fence: only ordered behind the scan
        store.fence(user.cuda_stream)
        with torch.cuda.stream(user):
            late.copy_(rows)                   # behind the fence: ordered behind the select
        user.synchronize()
        stale_seen += int(early.cpu().tolist() != want_rows.astype(np.int32).tolist())
        fenced_ok += int(late.cpu().tolist() == want_rows.astype(np.int32).tolist())
    assert fenced_ok == 8
    assert stale_seen >= 1, "the unfenced copy was never early: is the select still on the caller's stream?"
    if pipe_mode == 2:
        assert stale_seen == 8
    store.close()


def _device_query(store, dq, b, d, k, out, stream):
    rows, scores, cnt, status = out
    kk = rows.shape[1]
    store.search_device(dq.data_ptr() + b * d * 4, 1, k, rows.data_ptr() + b * kk * 4, scores.data_ptr() + b * kk * 4,
                        cnt.data_ptr() + b * 4, status.data_ptr() + b * 4, stream)


@pytest.mark.parametrize("d", [64, 10])
def test_deferred_selects_survive_whatever_comes_between_two_queries(trag, oracle, d):
    """Mode 2 keeps the select of the last query pending until the next scan's launch carries it. Between
    two queries: another k, inserts (the pending select belongs to the store as it was), a removal, growth
    past the reserved capacity (the scratch sets are replaced), a switch of the mode. Every result equals
    the oracle's on the corpus of its own moment. (d = 10: rows the fused launch does not take -- every
    select is launched on its own; same results.)"""
    import torch
    n, kmax = 40_000, 50
    store = trag.VectorStore(trag.VectorStoreConfig(dimension=d, device=0, reserve_rows=n + 100))
    store.fill_synthetic(n, 181)
    Q = oracle.synth_rows(182, 0, 8, d)
    dq = torch.from_numpy(Q).cuda()
    out = (torch.full((8, kmax), -1, dtype=torch.int32, device="cuda"), torch.zeros((8, kmax), device="cuda"),
           torch.zeros(8, dtype=torch.int32, device="cuda"), torch.full((8,), 9, dtype=torch.int32, device="cuda"))
    user = torch.cuda.Stream()
    sp = user.cuda_stream
    torch.cuda.synchronize()
    store.set_pipelined(2)
    corpus = {}
    ks = [10, 5, 50, 10, 10, 7, 10, 10]
    corpus[0] = store.read_rows(0, store.rows())
    _device_query(store, dq, 0, d, ks[0], out, sp)
    _device_query(store, dq, 1, d, ks[1], out, sp)              # another k, same corpus
    corpus[1] = corpus[0]
    extra = oracle.synth_rows(183, 0, 60, d)
    store.insert_rows(extra[:40])                               # small inserts: deferred themselves
    corpus[2] = store.read_rows(0, store.rows())
    _device_query(store, dq, 2, d, ks[2], out, sp)
    store.remove_row(int(oracle.search(corpus[2], Q[3], 1, 0)[0][0]))   # query 3's best row goes
    corpus[3] = store.read_rows(0, store.rows())
    _device_query(store, dq, 3, d, ks[3], out, sp)
    big = oracle.synth_rows(184, 0, 30_000, d)
    store.insert_rows(big)                                      # past the reservation: everything is replaced
    corpus[4] = store.read_rows(0, store.rows())
    _device_query(store, dq, 4, d, ks[4], out, sp)
    store.set_pipelined(0)                                      # the pending select is launched by the switch
    corpus[5] = corpus[4]
    _device_query(store, dq, 5, d, ks[5], out, sp)
    store.set_pipelined(2)
    corpus[6] = corpus[4]
    _device_query(store, dq, 6, d, ks[6], out, sp)
    got_host = store.search_rows(Q[7], ks[7])                   # host entry with a select pending
    store.fence(sp)
    user.synchronize()
    torch.cuda.synchronize()
    rows, scores, cnt, status = (t.cpu().numpy() for t in out)
    removed = int(oracle.search(corpus[2], Q[3], 1, 0)[0][0])

    def expect(b, c):
        dead = None
        if b >= 3:
            dead = np.zeros(c.shape[0], dtype=np.uint8)
            dead[removed] = 1
        return oracle.search(c, Q[b], ks[b], 0, dead)

    for b in range(7):
        want = expect(b, corpus[b])
        assert status[b] == 0 and cnt[b] == ks[b], (b, status[b], cnt[b])
        assert_same_ranking(rows[b, :ks[b]].astype(np.uint32), scores[b, :ks[b]], want[0], want[1], f"query {b}")
    want = expect(7, corpus[4])
    assert_same_ranking(got_host[0], got_host[1], want[0], want[1], "host entry")
    store.close()


def test_deferred_select_pending_at_destroy_is_dropped_quietly(trag, oracle):
    import torch
    d, k = 64, 10
    store = trag.VectorStore(trag.VectorStoreConfig(dimension=d, device=0))
    store.fill_synthetic(20_000, 185)
    dq = torch.from_numpy(oracle.synth_rows(186, 0, 1, d)).cuda()
    out = (torch.full((1, k), -1, dtype=torch.int32, device="cuda"), torch.zeros((1, k), device="cuda"),
           torch.zeros(1, dtype=torch.int32, device="cuda"), torch.full((1,), 9, dtype=torch.int32, device="cuda"))
    store.set_pipelined(2)
    _device_query(store, dq, 0, d, k, out, 0)
    store.close()                                                # no fence: the result is never written
    torch.cuda.synchronize()
    assert out[0].cpu().tolist() == [[-1] * k] and out[3].cpu().tolist() == [9]
