"""Sharded search behind the C ABI (trag_shard_group_*, trag_sharded_*) on the one GPU of the test
box: RCCL really runs (a communicator of one rank; its all-gather and broadcast are executed by
librccl), and the n > 1 logic -- striping, global row ids, the exchange, the merge, the tie rule
across shards -- runs with the same GPU named twice over the copy exchange, which RCCL itself
refuses. Every result is compared with the oracle's search over the whole corpus, bit for bit."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import has_gpu
from util import assert_same_ranking

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs a GPU")]


def test_rccl_is_loaded_and_reports_a_version(trag):
    from trueno_rag_amd.sharded import rccl_version
    assert rccl_version() >= 20000


@pytest.mark.parametrize("overlap", [False, True])
def test_shard_group_one_rank_runs_rccl(trag, oracle, overlap):
    """world = 1: pack -> ncclAllGather -> merge, enqueued by ONE C call; rows come back global."""
    import torch
    from trueno_rag_amd.sharded import ShardGroup
    n, d, k, B = 200_000, 64, 10, 5
    store = trag.VectorStore(trag.VectorStoreConfig(dimension=d, device=0, reserve_rows=n))
    store.fill_synthetic(n, 31)
    corpus = store.read_rows(0, n)
    Q = oracle.synth_rows(32, 0, B, d)
    g = ShardGroup(0, 1, 0, k_max=100, max_queries=8, overlap=overlap)
    dq = torch.from_numpy(Q).cuda()
    row0 = 1_000_000   # this "shard" owns global rows [1 000 000, 1 200 000)
    stream = torch.cuda.current_stream().cuda_stream
    for rep in range(3):   # three calls in a row: both buffer slots and their reuse
        rows = torch.zeros((B, k), dtype=torch.int32, device="cuda")
        scores = torch.zeros((B, k), dtype=torch.float32, device="cuda")
        cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
        status = torch.full((B,), 9, dtype=torch.int32, device="cuda")
        g.search_device(store, row0, 1, dq.data_ptr(), B, k, rows.data_ptr(), scores.data_ptr(), cnt.data_ptr(),
                        status.data_ptr(), stream)
        g.fence(stream)
        torch.cuda.synchronize()
        assert status.cpu().tolist() == [0] * B and cnt.cpu().tolist() == [k] * B
        for b in range(B):
            er, es = oracle.search(corpus, Q[b], k, 0)
            assert_same_ranking(rows[b].cpu().numpy().astype(np.uint32), scores[b].cpu().numpy(),
                                er + row0, es, f"rep {rep} query {b}")
    # host entry: any batch size (chunked by max_queries), synchronous, striped row mapping
    Q2 = oracle.synth_rows(33, 0, 19, d)
    r, s, c = g.search_batch_rows(store, 3, 7, Q2, 25)     # global row = 3 + 7 * local
    for b in range(19):
        er, es = oracle.search(corpus, Q2[b], 25, 0)
        assert_same_ranking(r[b, :c[b]], s[b, :c[b]], er * 7 + 3, es, f"host entry query {b}")
    # limits are errors, not faults
    with pytest.raises(trag.TragError):
        g.search_device(store, 0, 1, dq.data_ptr(), 9, k, rows.data_ptr(), scores.data_ptr(), cnt.data_ptr())
    with pytest.raises(trag.TragError):
        g.search_device(store, 0, 1, dq.data_ptr(), 1, 101, rows.data_ptr(), scores.data_ptr(), cnt.data_ptr())
    g.close()
    store.close()


@pytest.mark.parametrize("mirror", [False, True])
@pytest.mark.parametrize("metric", [0, 1])
def test_overlap_mode_single_queries_exchange_one_call_late(trag, oracle, metric, mirror):
    """Overlap mode, one query per call: the local store defers each select into its next scan's launch,
    and the group enqueues the exchange of call i when call i + 1 has been enqueued (or at the fence).
    150 calls back to back, nothing synchronised (more than two laps of the 64-entry result ring, several
    of its periodic waits), k changing from call to call, a 5-query call and a host-entry call in the
    middle (both settle what is pending first), a second stream near the end: after one fence every
    call's output holds the oracle's global rows and score bits. (The Euclidean store is short: its plan is the
    multi-level select, whose stages the store runs on its own side stream -- nothing is deferred, and the
    exchange must wait for that stream; the same with a bf16 mirror, whose scan's last stages run there.)"""
    import torch
    from trueno_rag_amd.sharded import ShardGroup
    n, d, N = 120_000, 64, 150
    store = trag.VectorStore(trag.VectorStoreConfig(dimension=d, metric=trag.DistanceMetric(metric), device=0,
                                                    reserve_rows=n))
    store.fill_synthetic(n, 41)
    if mirror:
        store.set_bf16_mirror(True)
    corpus = store.read_rows(0, n)
    Q = oracle.synth_rows(42, 0, N, d)
    dq = torch.from_numpy(Q).cuda()
    g = ShardGroup(0, 1, 0, k_max=20, max_queries=8, overlap=True)
    ks = [(3, 10, 20, 7)[i % 4] for i in range(N)]
    kmax, row0 = 20, 5_000_000
    rows = torch.full((N, kmax), -1, dtype=torch.int32, device="cuda")
    scores = torch.zeros((N, kmax), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(N, dtype=torch.int32, device="cuda")
    status = torch.full((N,), 9, dtype=torch.int32, device="cuda")
    user, other = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    mid = {}
    for i in range(N):
        st = other if 140 <= i < 143 else user
        if i == 70:      # a batch of 5 through the same group: queries 70..74 in one call
            g.search_device(store, row0, 1, dq.data_ptr() + i * d * 4, 5, 10, rows.data_ptr() + i * kmax * 4,
                            scores.data_ptr() + i * kmax * 4, cnt.data_ptr() + i * 4, status.data_ptr() + i * 4,
                            st.cuda_stream)
            continue
        if 70 < i < 75:
            continue
        if i == 100:
            mid["host"] = g.search_batch_rows(store, row0, 1, Q[100:101], 10)
        g.search_device(store, row0, 1, dq.data_ptr() + i * d * 4, 1, ks[i], rows.data_ptr() + i * kmax * 4,
                        scores.data_ptr() + i * kmax * 4, cnt.data_ptr() + i * 4, status.data_ptr() + i * 4,
                        st.cuda_stream)
    g.fence(user.cuda_stream)
    user.synchronize()
    torch.cuda.synchronize()
    R, S, C, T = rows.cpu().numpy(), scores.cpu().numpy(), cnt.cpu().numpy(), status.cpu().numpy()
    for i in range(N):
        if 70 <= i < 75:
            # (a batch call writes its outputs k apart: query b of the call at offset b * 10 of the call's block)
            b = i - 70
            flat_r = R[70:75].reshape(-1)[b * 10:(b + 1) * 10]
            flat_s = S[70:75].reshape(-1)[b * 10:(b + 1) * 10]
            er, es = oracle.search(corpus, Q[i], 10, metric)
            assert_same_ranking(flat_r.astype(np.uint32), flat_s, er + row0, es, f"batch call, query {i}")
            continue
        k = ks[i]
        er, es = oracle.search(corpus, Q[i], k, metric)
        assert T[i] == 0 and C[i] == k, (i, T[i], C[i])
        assert_same_ranking(R[i, :k].astype(np.uint32), S[i, :k], er + row0, es, f"call {i}")
    er, es = oracle.search(corpus, Q[100], 10, metric)
    assert_same_ranking(mid["host"][0][0], mid["host"][1][0], er + row0, es, "host entry in the middle")
    g.close()
    store.close()


@pytest.mark.parametrize("pipe_mode", [1, 2])
def test_group_without_overlap_on_a_store_its_caller_pipelined(trag, oracle, pipe_mode):
    """A group created without overlap enqueues the exchange on the caller's stream right behind the local
    search. If the CALLER has put the local store into a pipelined mode, that search's select is on the
    store's side stream or still pending: the group fences the store first. Six single-query calls and a
    batch of three, one stream, one synchronisation."""
    import torch
    from trueno_rag_amd.sharded import ShardGroup
    n, d, k, row0 = 90_000, 64, 10, 1_000_000
    store = trag.VectorStore(trag.VectorStoreConfig(dimension=d, device=0, reserve_rows=n))
    store.fill_synthetic(n, 51)
    store.set_pipelined(pipe_mode)
    corpus = store.read_rows(0, n)
    Q = oracle.synth_rows(52, 0, 9, d)
    dq = torch.from_numpy(Q).cuda()
    g = ShardGroup(0, 1, 0, k_max=k, max_queries=8, overlap=False)
    rows = torch.full((9, k), -1, dtype=torch.int32, device="cuda")
    scores = torch.zeros((9, k), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(9, dtype=torch.int32, device="cuda")
    status = torch.full((9,), 9, dtype=torch.int32, device="cuda")
    user = torch.cuda.Stream()
    torch.cuda.synchronize()
    for i in list(range(6)) + [6]:
        nq = 1 if i < 6 else 3
        g.search_device(store, row0, 1, dq.data_ptr() + i * d * 4, nq, k, rows.data_ptr() + i * k * 4,
                        scores.data_ptr() + i * k * 4, cnt.data_ptr() + i * 4, status.data_ptr() + i * 4, user.cuda_stream)
    user.synchronize()
    R, S, C, T = rows.cpu().numpy(), scores.cpu().numpy(), cnt.cpu().numpy(), status.cpu().numpy()
    for i in range(9):
        er, es = oracle.search(corpus, Q[i], k, 0)
        assert T[i] == 0 and C[i] == k, (i, T[i], C[i])
        assert_same_ranking(R[i].astype(np.uint32), S[i], er + row0, es, f"call {i}")
    g.close()
    store.close()


def _global_oracle(oracle, corpus, dead, q, k, metric):
    d8 = np.zeros(corpus.shape[0], dtype=np.uint8)
    d8[list(dead)] = 1
    return oracle.search(corpus, q, k, metric, dead=d8)


def test_multi_gpu_store_one_device_over_rccl(trag, oracle):
    """n_devices = 1: ncclCommInitAll, the query broadcast is skipped, ncclAllGather runs."""
    from trueno_rag_amd.sharded import MultiGpuVectorStore
    n, d = 70_000, 96
    ms = MultiGpuVectorStore(trag.VectorStoreConfig(dimension=d, reserve_rows=n), [0], exchange="rccl")
    corpus = oracle.synth_rows(77, 0, n, d)
    assert ms.insert_rows(corpus[:50_000]) == 0 and ms.insert_rows(corpus[50_000:]) == 50_000
    ms.remove_row(123)
    Q = oracle.synth_rows(78, 0, 12, d)
    r, s, c = ms.search_batch_rows(Q, 10)
    for b in range(12):
        er, es = _global_oracle(oracle, corpus, {123}, Q[b], 10, 0)
        assert_same_ranking(r[b, :c[b]], s[b, :c[b]], er, es, f"query {b}")
    assert ms.len() == n - 1 and ms.rows() == n
    ms.close()


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_multi_gpu_store_striping_exchange_and_ties(trag, oracle, metric):
    """Three shards on the one GPU (copy exchange): ragged appends, duplicates that tie across
    shards, removals, overwrites, single queries and a batch, k beyond a shard's and beyond the
    store's rows."""
    from trueno_rag_amd.sharded import MultiGpuVectorStore
    d = 64
    rng = np.random.default_rng(5 + metric)
    ms = MultiGpuVectorStore(trag.VectorStoreConfig(dimension=d, metric=trag.DistanceMetric(metric)), [0, 0, 0],
                             exchange="copy")
    base = oracle.synth_rows(500 + metric, 0, 9000, d)
    base[4000:4040] = base[17]            # 40 copies of row 17, spread over the three shards
    base[8000] = 0.0                      # a zero row
    blocks, at = [1, 7, 1000, 3, 5000, 2989], 0
    for nb in blocks:
        assert ms.insert_rows(base[at:at + nb]) == at
        at += nb
    assert at == 9000 and ms.rows() == 9000
    dead = {17, 4001, 4002, 8999, 0}
    for r in dead:
        ms.remove_row(r)
    new_row = oracle.synth_rows(900, 0, 1, d)[0]
    ms.overwrite_row(5555, new_row)
    base[5555] = new_row
    assert ms.len() == 9000 - len(dead)
    Q = np.concatenate([oracle.synth_rows(600 + metric, 0, 30, d), base[17:18], np.zeros((1, d), np.float32)])
    for k in (1, 10, 64, 200):
        r, s, c = ms.search_batch_rows(Q, k)
        for b in range(Q.shape[0]):
            er, es = _global_oracle(oracle, base, dead, Q[b], k, metric)
            assert c[b] == len(er)
            assert_same_ranking(r[b, :c[b]], s[b, :c[b]], er, es, f"metric {metric} k {k} query {b}")
    r1, s1 = ms.search_rows(Q[3], 10)
    er, es = _global_oracle(oracle, base, dead, Q[3], 10, metric)
    assert_same_ranking(r1, s1, er, es, "single query")
    # k beyond the whole store: everything comes back, in order
    small = MultiGpuVectorStore(trag.VectorStoreConfig(dimension=d, metric=trag.DistanceMetric(metric)), [0, 0],
                                exchange="copy")
    small.insert_rows(base[:5])
    r, s = small.search_rows(Q[0], 2 ** 31)
    er, es = oracle.search(base[:5], Q[0], 5, metric)
    assert_same_ranking(r, s, er, es, "k = 2^31 on five rows")
    empty_r, empty_s = MultiGpuVectorStore(trag.VectorStoreConfig(dimension=d), [0, 0], exchange="copy").search_rows(Q[0], 3)
    assert len(empty_r) == 0
    with pytest.raises(trag.DimensionMismatch):
        ms.insert_rows(np.zeros((2, d + 1), np.float32))
    small.close()
    ms.close()


def test_multi_gpu_fill_synthetic_is_the_global_generator(trag, oracle):
    """Striped synthetic fill: shard j's local row r is generator row r * G + j."""
    from trueno_rag_amd.sharded import MultiGpuVectorStore
    d, n = 32, 10_001
    ms = MultiGpuVectorStore(trag.VectorStoreConfig(dimension=d), [0, 0, 0, 0], exchange="copy")
    assert ms.fill_synthetic(4000, 99) == 0 and ms.fill_synthetic(n - 4000, 99) == 4000
    corpus = oracle.synth_rows(99, 0, n, d)
    Q = oracle.synth_rows(98, 0, 4, d)
    r, s, c = ms.search_batch_rows(Q, 20)
    for b in range(4):
        er, es = oracle.search(corpus, Q[b], 20, 0)
        assert_same_ranking(r[b], s[b], er, es, f"query {b}")
    ms.close()


def test_torch_nccl_backend_one_rank_exchange():
    """The Python ShardExchange over torch.distributed's nccl backend (= RCCL) with a group of one
    rank: all_gather_into_tensor on device records really runs through RCCL."""
    code = textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {ROOT!r}); sys.path.insert(0, os.path.join({ROOT!r}, "tests"))
        import numpy as np, torch, torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 400))
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        from conftest import load_package
        pkg = load_package()
        from oracle import oracle
        from trueno_rag_amd.sharded import ShardExchange
        n, d, k, B = 50_000, 64, 10, 4
        store = pkg.VectorStore(pkg.VectorStoreConfig(dimension=d, device=0, reserve_rows=n))
        store.fill_synthetic(n, 11)
        corpus = store.read_rows(0, n)
        Q = oracle.synth_rows(12, 0, B, d)
        dq = torch.from_numpy(Q).cuda()
        rows = torch.zeros((B, k), dtype=torch.int32, device="cuda"); scores = torch.zeros((B, k), device="cuda")
        cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        store.search_device(dq.data_ptr(), B, k, rows.data_ptr(), scores.data_ptr(), cnt.data_ptr(), 0, st)
        ex = ShardExchange(k, max_queries=B, device=torch.device("cuda", 0))
        o_r = torch.zeros_like(rows); o_s = torch.zeros_like(scores); o_n = torch.zeros_like(cnt)
        ex.exchange(rows, scores, cnt, 500, o_r, o_s, o_n, st)
        torch.cuda.synchronize()
        for b in range(B):
            er, es = oracle.search(corpus, Q[b], k, 0)
            assert o_r[b].cpu().numpy().tolist() == (er + 500).tolist(), b
            assert o_s[b].cpu().numpy().tobytes() == es.tobytes(), b
        dist.destroy_process_group()
        print("NCCL_ONE_RANK_OK")
    """)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=170, cwd=ROOT)
    assert out.returncode == 0 and "NCCL_ONE_RANK_OK" in out.stdout, out.stderr[-3000:]
