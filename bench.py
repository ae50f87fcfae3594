#!/usr/bin/env python3
"""Headline benchmark: brute-force cosine top-10 over an N x d fp32 corpus
resident in HBM, one query per step, through the C ABI (libtrag.so).

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1: the fixed corpus is row-sharded over the ranks (one process per GPU);
every step each rank scans its shard, the per-shard top-k lists are exchanged
with one RCCL all-gather and merged on every rank (strong scaling).
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=10_000_000, help="corpus rows (whole job)")
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--metric", choices=["cosine", "euclidean", "dot"], default="cosine",
                    help="DistanceMetric of the store (the headline is cosine, the reference's default)")
    ap.add_argument("--mode", type=int, default=0, help="0 auto, 1 exact scan, 2 streaming scan, 3 f32 matrix cores, 4 bf16 prefilter + exact finish, 5 bf16 mirror")
    ap.add_argument("--mirror", action="store_true",
                    help="keep a bf16 copy of the corpus in HBM (1.5x memory): prefilter passes read half the bytes, "
                         "results stay the exact f32 results")
    ap.add_argument("--batch", type=int, default=1, help="queries per step (2..8: one multi-query exact pass; above: matrix-core path)")
    ap.add_argument("--sparse-vocab", type=int, default=50000)
    ap.add_argument("--sparse-tokens", type=int, default=64)
    ap.add_argument("--sparse-terms", type=int, default=8)
    ap.add_argument("--hybrid", action="store_true",
                    help="hybrid step: dense top-50 + BM25 top-50 (synthetic documents) -> RRF(60) -> top-k on the host")
    ap.add_argument("--host", action="store_true",
                    help="time the host-pointer entry (trag_store_search: query from host memory, results to host "
                         "memory, synchronous) instead of the device-resident one; the shape of the smallest benchmark configuration (10 000 x 384)")
    ap.add_argument("--pipeline", type=int, choices=(1, 2), default=None,
                    help="pipelined mode of single-query steps: 2 (default) deferred selects, 1 side stream")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="single-query steps: keep the select on the stream of the scans (by default the handle is in "
                         "pipelined mode: the select of query i runs beside the scan of query i+1, results fenced)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the rocprofv3 counter pass that measures the scan kernel's HBM traffic")
    ap.add_argument("--cpu-sample-rows", type=int, default=1_000_000)
    return ap.parse_args()


def unit_queries(n, d, seed):
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((n, d)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True).astype(np.float32)
    return np.ascontiguousarray(q, dtype=np.float32)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # (inside a profiler session the preloaded library has already initialised the GPU in this
    # process, and a GPU-initialised process must not start other programs)
    profiled = "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(
        n.startswith(("ROCPROF", "ROCP_")) for n in os.environ)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        if profiled:
            sys.exit("bench.py --gpus N under a profiler: this process may not start the ranks (the profiler has "
                     "initialised the GPU in it). Profile one rank (--gpus 1 --rows <rows per GPU>), or start the "
                     "ranks with torch.distributed.run outside the profiler.")
        # started as plain `python bench.py --gpus N`: launch the N ranks (one process per GPU) as
        # children of this process, which has not touched the GPU, and hand their exit code on
        import socket
        import subprocess
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    if world != args.gpus and rank == 0:
        print(f"--gpus {args.gpus} but WORLD_SIZE={world}: running with {world} rank(s)", file=sys.stderr)
    # stdout carries ONE line, the result. Whatever libraries write to file descriptor 1 on their own (RCCL prints a
    # version banner there when NCCL_DEBUG is set in the environment) goes to stderr instead.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    import torch.distributed as dist
    # HBM traffic of the scan kernel from the hardware counters: a short copy of this very run
    # under rocprofv3, started before this process touches the GPU
    traffic = None   # (never from inside a profiler session, see above)
    if (world == 1 and not args.no_pmc and not args.hybrid and not args.host and not profiled
            and not os.environ.get("TRAG_BENCH_CHILD")):
        traffic = pmc_traffic(args)
    # TRAG_BENCH_ONE_GPU=1: rehearsal of the N > 1 code path on a one-GPU box — every rank uses
    # cuda:0 and the collectives go through gloo (RCCL refuses two ranks on one device). The
    # numbers of such a run mean nothing; it exists to exercise the sharded step end to end.
    one_gpu = bool(os.environ.get("TRAG_BENCH_ONE_GPU"))
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # torch.distributed is the control plane only (rendezvous, barrier, the max over ranks of
        # the elapsed time: host memory over gloo); the data-path collective -- the all-gather of the
        # per-shard top-k lists -- is RCCL inside libtrag.so (trag_shard_search_device)
        dist.init_process_group("gloo")
    pkg = load_package()
    L = pkg.lib()

    # A rank stuck in the communicator's set-up or in a collective must not sit there until somebody's
    # timeout: with more than one rank (or TRAG_BENCH_DEBUG) every stage of the run re-arms a watchdog
    # that prints the stack of the stuck rank -- the line it is stuck on -- and exits non-zero.
    watchdog_s = float(os.environ.get("TRAG_BENCH_WATCHDOG_S", "300"))
    watched = world > 1 or bool(os.environ.get("TRAG_BENCH_DEBUG"))
    if watched:
        import faulthandler

    def stage(msg):
        if watched:
            faulthandler.dump_traceback_later(watchdog_s, exit=True)
        if os.environ.get("TRAG_BENCH_DEBUG"):
            print(f"[debug] rank {rank}: {msg}", file=sys.stderr, flush=True)

    stage("process group up")
    n_total, d, k = args.rows, args.dim, args.k
    per = (n_total + world - 1) // world
    row0 = rank * per
    n_local = max(0, min(per, n_total - row0))
    seed = 1004
    metric_id = {"cosine": 0, "euclidean": 1, "dot": 2}[args.metric]
    store = pkg.VectorStore(pkg.VectorStoreConfig(dimension=d, metric=pkg.DistanceMetric(metric_id), device=local,
                                                  scan_mode=args.mode, reserve_rows=n_local))
    store.fill_synthetic(n_local, seed, first_global_row=row0)
    if args.mirror:
        store.set_bf16_mirror(True)
    pipelined = not args.no_pipeline and args.batch == 1 and not args.host and not args.hybrid
    # 2: the select of query i rides in the launch of query i + 1's scan (one launch a query); 1: it runs
    # on the handle's side stream beside that scan -- what a sharded run's exchange group switches on
    # itself (overlap mode: the exchange needs each query's results while the next scan runs)
    exchanging = world > 1 or bool(os.environ.get("TRAG_BENCH_FORCE_EXCHANGE"))
    pipe_mode = (args.pipeline if args.pipeline is not None else (1 if exchanging else 2)) if pipelined else 0
    if pipelined:
        store.set_pipelined(pipe_mode)

    B = args.batch
    n_q = (args.steps + args.warmup) * B
    Q = torch.from_numpy(unit_queries(n_q, d, seed + 100)).cuda()
    out_rows = torch.zeros((n_q, k), dtype=torch.int32, device="cuda")
    out_scores = torch.zeros((n_q, k), dtype=torch.float32, device="cuda")
    out_n = torch.zeros(n_q, dtype=torch.int32, device="cuda")
    status = torch.zeros(n_q, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream()
    sptr = stream.cuda_stream
    esz = 4
    ex = None      # rehearsal only: the Python exchange over gloo (several ranks on one GPU)
    group = None   # the real thing: one C call per step, RCCL inside the library
    # TRAG_BENCH_FORCE_EXCHANGE=1 runs the pack / gather / merge leg on one GPU too (world size 1)
    if world > 1 or os.environ.get("TRAG_BENCH_FORCE_EXCHANGE"):
        from trueno_rag_amd.sharded import ShardExchange, ShardGroup
        BX = max(B, 1000)   # the second leg of an N > 1 run sends 1000 queries per step through the exchange
        if one_gpu:
            ex = ShardExchange(k, max_queries=BX, device=torch.device("cuda", local))
            side = torch.cuda.Stream()
        else:
            # overlap: pack / all-gather / merge of step i run on the library's exchange stream while
            # the scan of step i+1 runs on ours; group.fence() before the results are consumed
            group = ShardGroup(rank, world, local, k_max=k, max_queries=BX, overlap=True)
        m_rows = torch.zeros((n_q, k), dtype=torch.int32, device="cuda")
        m_scores = torch.zeros((n_q, k), dtype=torch.float32, device="cuda")
        m_n = torch.zeros(n_q, dtype=torch.int32, device="cuda")

    sparse_info = None
    if args.hybrid:
        assert world == 1 and B == 1, "--hybrid is a single-GPU, single-query step"
        cand = 50  # HybridRetrieverConfig::candidates_per_source, src/retrieve.rs:94
        fusion = pkg.FusionStrategy.RRF(60.0)
        import importlib
        rrf_device = importlib.import_module(pkg.__name__ + ".fusion").rrf_device
        h_rows = torch.zeros((n_q, cand), dtype=torch.int32, device="cuda")
        h_scores = torch.zeros((n_q, cand), dtype=torch.float32, device="cuda")
        s_docs = torch.zeros((n_q, cand), dtype=torch.int32, device="cuda")
        s_scores = torch.zeros((n_q, cand), dtype=torch.float32, device="cuda")
        s_n = torch.zeros(n_q, dtype=torch.int32, device="cuda")
        s_st = torch.zeros(n_q, dtype=torch.int32, device="cuda")
        f_ids = torch.zeros((n_q, k), dtype=torch.int32, device="cuda")
        f_scores = torch.zeros((n_q, k), dtype=torch.float32, device="cuda")
        f_n = torch.zeros(n_q, dtype=torch.int32, device="cuda")
        bm_stream = torch.cuda.Stream()
        bm_done = torch.cuda.Event()
        # the sparse half: one synthetic document per corpus row (Zipf(1.0) over the vocabulary),
        # queries drawn from the same law
        bm25 = pkg.BM25Index(device=local)
        t_build = time.perf_counter()
        bm25.fill_synthetic(n_local, args.sparse_vocab, args.sparse_tokens, seed + 1, first_global_doc=row0)
        bm25.search_ids([0], 1)  # builds and uploads the inverted lists
        t_build = time.perf_counter() - t_build
        q_terms = sparse_queries(n_q, args.sparse_vocab, args.sparse_terms, seed + 101)
        for j in range(min(4, n_q)):   # the host entry learns the select plan this corpus needs (crowded score
            bm25.search_ids(q_terms[j], cand)   # bins at millions of documents); the device entry then uses it
        sparse_info = {"documents": n_local, "vocab": args.sparse_vocab, "tokens_per_doc": args.sparse_tokens,
                       "query_terms": args.sparse_terms, "postings": bm25.stats()["postings"],
                       "build_seconds": t_build}

    def hybrid_step(i):
        """src/retrieve.rs:175-220 without a host round trip: dense top-50 on the bench's stream, BM25
        top-50 on a second stream beside it, then RRF(60) and the cut to k on the device
        (trag_fuse_rrf_device), ordered behind both by an event. Nothing is synchronised per step."""
        store.search_device(Q.data_ptr() + i * d * esz, 1, cand, h_rows.data_ptr() + i * cand * esz,
                            h_scores.data_ptr() + i * cand * esz, out_n.data_ptr() + i * esz,
                            status.data_ptr() + i * esz, sptr)
        bm25.search_ids_device(q_terms[i], cand, s_docs.data_ptr() + i * cand * esz, s_scores.data_ptr() + i * cand * esz,
                               s_n.data_ptr() + i * esz, s_st.data_ptr() + i * esz, bm_stream.cuda_stream)
        bm_done.record(bm_stream)
        stream.wait_event(bm_done)
        rrf_device(h_rows.data_ptr() + i * cand * esz, out_n.data_ptr() + i * esz, cand,
                   s_docs.data_ptr() + i * cand * esz, s_n.data_ptr() + i * esz, cand, 1, 60.0, k,
                   f_ids.data_ptr() + i * k * esz, f_scores.data_ptr() + i * k * esz, f_n.data_ptr() + i * esz, sptr)

    Q_host = Q.cpu().numpy() if args.host else None
    host_out = {}

    def step(s):
        if args.hybrid:
            return hybrid_step(s)
        i = s * B
        if args.host:   # PCIe and the synchronisation are inside the step here, by definition of this mode
            host_out["rows"], host_out["scores"], _ = store.search_batch_rows(Q_host[i:i + B], k)
            return
        if group is not None:   # local scan -> records -> ONE RCCL all-gather -> merge: one C call
            group.search_device(store, row0, 1, Q.data_ptr() + i * d * esz, B, k, m_rows.data_ptr() + i * k * esz,
                                m_scores.data_ptr() + i * k * esz, m_n.data_ptr() + i * esz,
                                status.data_ptr() + i * esz, sptr)
            return
        store.search_device(Q.data_ptr() + i * d * esz, B, k, out_rows.data_ptr() + i * k * esz,
                            out_scores.data_ptr() + i * k * esz, out_n.data_ptr() + i * esz,
                            status.data_ptr() + i * esz, sptr)
        if ex is not None:  # pack, ONE all-gather for the step's queries, device merge
            done = torch.cuda.Event()
            done.record(stream)
            side.wait_event(done)
            store.fence(side.cuda_stream)   # (a pipelined handle's select is not on `stream`)
            with torch.cuda.stream(side):
                ex.exchange(out_rows[i:i + B], out_scores[i:i + B], out_n[i:i + B], row0,
                            m_rows[i:i + B], m_scores[i:i + B], m_n[i:i + B], side.cuda_stream)

    def fence():
        store.fence(sptr)          # pipelined handle: the selects on its side stream
        if group is not None:
            group.fence(sptr)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    stage("store filled, buffers allocated")
    for i in range(args.warmup):
        step(i)
        stage(f"warmup step {i} enqueued")
    fence()
    stage("warmup done")
    # an event pair costs ~6 us of stream time: on sub-millisecond scans time every 4th launch only
    prof_every = 4 if args.steps >= 40 else 1
    # (hybrid: BM25 runs beside the scan on a second stream, and an event pair around the scan then also
    # spans whatever of the previous step is still draining: the scan is timed alone, after the steps)
    store.set_profiling(0 if (os.environ.get("TRAG_BENCH_NO_PROFILE") or args.hybrid) else prof_every)
    import gc
    gc.collect()
    gc.disable()   # as timeit does: a generation-2 collection of this process's heap is a ~70 ms pause
    t0 = time.perf_counter()
    dbg = []
    for i in range(args.warmup, args.warmup + args.steps):
        step(i)
        if os.environ.get("TRAG_BENCH_DEBUG"):
            dbg.append(time.perf_counter())
    if dbg:
        print("[debug] step ms:", [round(1e3 * (b - a), 2) for a, b in zip([t0] + dbg, dbg)], file=sys.stderr)
    t_loop = time.perf_counter() - t0
    fence()
    elapsed = time.perf_counter() - t0
    gc.enable()
    stage("timed steps done")
    if os.environ.get("TRAG_BENCH_DEBUG"):
        print(f"[debug] loop {1e3 * t_loop:.2f} ms, fence {1e3 * (elapsed - t_loop):.2f} ms", file=sys.stderr)
    if args.hybrid and not os.environ.get("TRAG_BENCH_NO_PROFILE"):
        # the dense scan of the last steps' queries once more, alone on the stream, every launch timed
        prof_every = 1
        store.set_profiling(1)
        for i in range(max(args.warmup, args.warmup + args.steps - 10), args.warmup + args.steps):
            store.search_device(Q.data_ptr() + i * d * esz, 1, cand, h_rows.data_ptr() + i * cand * esz,
                                h_scores.data_ptr() + i * cand * esz, out_n.data_ptr() + i * esz,
                                status.data_ptr() + i * esz, sptr)
        torch.cuda.synchronize()
    scan_ms, scan_launches = store.scan_time()
    store.set_profiling(False)
    mode_used = store.last_stats()["mode_used"]  # the path the timed steps actually took
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    batch_leg, batch_sample = None, None
    if (group is not None or ex is not None) and not args.hybrid and B != 1000:
        stage("second leg: 1000 queries per step")
        batch_leg, batch_sample = exchange_batch_leg(args, store, group, ex, side if ex is not None else None, rank,
                                                     world, row0, d, k, seed, sptr, stream, fence, dist)
    # what RCCL itself says about the communicator of every rank (evidence that the exchange ran over
    # `world` ranks, one GPU each)
    ranks_info = None
    if group is not None:
        try:
            mine_info = dict(group.info(), rank=rank, local_device=local)
        except Exception as e:  # noqa: BLE001 -- an RCCL without the query calls: say so in the record
            mine_info = {"rank": rank, "local_device": local, "error": str(e)}
        props = torch.cuda.get_device_properties(local)
        mine_info.update({"host": os.uname().nodename, "pid": os.getpid(), "gpu": props.name,
                          "pci_bus_id": getattr(props, "pci_bus_id", None),
                          "gpu_uuid": str(getattr(props, "uuid", "")) or None})
        ranks_info = [mine_info]
        if world > 1:
            ranks_info = [None] * world
            dist.all_gather_object(ranks_info, mine_info)
    stage("parity checks")

    # ---- parity at full size (outside the timed region) ----
    uncertified = int((status[args.warmup * B:] != 0).sum().item())
    if args.hybrid:
        out_rows, out_scores = h_rows, h_scores
        k = cand
    sharded = ex is not None or group is not None
    if group is not None:   # the local list of the last query, for the host-side check of the merge
        store.search_device(Q.data_ptr() + (n_q - 1) * d * esz, 1, k, out_rows.data_ptr() + (n_q - 1) * k * esz,
                            out_scores.data_ptr() + (n_q - 1) * k * esz, out_n.data_ptr() + (n_q - 1) * esz, 0, sptr)
        torch.cuda.synchronize()
    final_rows = (m_rows if sharded else out_rows)[-1].cpu().numpy().astype(np.uint32)
    final_scores = (m_scores if sharded else out_scores)[-1].cpu().numpy()
    if args.host:
        final_rows, final_scores = host_out["rows"][-1].astype(np.uint32), host_out["scores"][-1]
        uncertified = 0   # the host entry repeats unfinished queries itself
    parity = {"uncertified_queries": uncertified, "sorted": bool(np.all(final_scores[:-1] >= final_scores[1:]))}
    if args.hybrid:
        # the sparse half finished on the device for every timed query, and the device fusion of the last
        # query equals the host fusion (trag_fuse) of the same two lists; its sparse list equals the host entry's
        parity["sparse_unfinished_queries"] = int((s_st[args.warmup:] != 0).sum().item())
        i = n_q - 1
        dn, sn = int(out_n[i].item()), int(s_n[i].item())
        dense = list(zip(h_rows[i, :dn].cpu().numpy().tolist(), h_scores[i, :dn].cpu().numpy().tolist()))
        sparse = list(zip(s_docs[i, :sn].cpu().numpy().tolist(), s_scores[i, :sn].cpu().numpy().tolist()))
        want = fusion.fuse(dense, sparse)[:args.k]   # (k is the candidate count from here on)
        fn = int(f_n[i].item())
        got_ids, got_scores = f_ids[i, :fn].cpu().numpy().tolist(), f_scores[i, :fn].cpu().numpy()
        parity["device_fusion_equals_host_fusion"] = bool(
            got_ids == [int(x) for x, _ in want]
            and got_scores.tobytes() == np.asarray([sc for _, sc in want], dtype=np.float32).tobytes())
        hd, hs = bm25.search_ids(q_terms[i], cand)
        parity["sparse_device_equals_host_entry"] = bool(
            hd.tolist() == [x for x, _ in sparse] and hs.tobytes() == s_scores[i, :sn].cpu().numpy().tobytes())
    if world > 1 or sharded:
        # the device merge of the last query against a host merge of every rank's own list
        # (descending score, ascending global row; gathered as plain Python objects)
        mine = (out_rows[-1].cpu().numpy().astype(np.int64) + row0, out_scores[-1].cpu().numpy(),
                int(out_n[-1].item()))
        lists = [None] * world
        if world > 1:
            dist.all_gather_object(lists, mine)
        else:
            lists = [mine]
        g_rows = np.concatenate([r[:n] for r, _, n in lists])
        g_scores = np.concatenate([sc[:n] for _, sc, n in lists])
        order = np.lexsort((g_rows, -g_scores.astype(np.float64)))[:k]
        parity["device_merge_equals_host_merge"] = bool(
            np.array_equal(g_rows[order], final_rows.astype(np.int64)[:len(order)])
            and g_scores[order].tobytes() == final_scores[:len(order)].tobytes())
    if sharded and not args.hybrid and rank == 0:
        # The sharded result against ONE store that holds the whole corpus on this rank's GPU (exact-order
        # scan: another route than the shards' automatic one), bit for bit: the last queries of the
        # headline leg and a sample of the 1000-query leg. And, as on the single-GPU line, the library
        # against the CPU oracle on the first rows of the corpus.
        nv = min(8, args.steps * B)
        head = (Q[n_q - nv:].cpu().numpy(), m_rows[n_q - nv:].cpu().numpy().astype(np.uint32),
                m_scores[n_q - nv:].cpu().numpy(), m_n[n_q - nv:].cpu().numpy(), status[n_q - nv:].cpu().numpy())
        parity.update(single_store_check(pkg, local, n_total, d, k, seed, metric_id, head, batch_sample,
                                         args.cpu_sample_rows, with_oracle=not args.no_cpu_baseline))
    if world == 1 and not sharded:
        # the exact-order scan of the same query must return the same rows and score bits
        store.set_scan_mode(1)
        er, es = store.search_rows(Q[-1].cpu().numpy(), k)
        # (long single-query scans skip the key stores no later stage reads; nonzero = the bound in force)
        parity["exact_scan_keys_kept_from"] = int(store.last_stats()["keys_kept_from"])
        # ... and an independent route over the full corpus: the streaming scan (another kernel, every
        # key stored, candidates re-scored in the reference's order) must agree bit for bit as well
        store.set_scan_mode(2)
        sr, ss = store.search_rows(Q[-1].cpu().numpy(), k)
        parity["streaming_scan_identical"] = bool(np.array_equal(sr, er) and ss.tobytes() == es.tobytes())
        store.set_scan_mode(args.mode)
        parity["exact_scan_identical"] = bool(np.array_equal(er, final_rows) and es.tobytes() == final_scores.tobytes())

    result = None
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        scan_avg_ms = scan_ms / max(scan_launches, 1)
        # bytes of corpus one scan launch must read: the f32 rows, or the bf16 mirror's when the pass reads that
        from_mirror = args.mirror and mode_used in (4, 5)
        algo_bytes = float(n_local) * d * (2.0 if from_mirror else 4.0)
        achieved = algo_bytes / (scan_avg_ms * 1e-3) / 1e9 if scan_avg_ms > 0 else 0.0
        # corpus passes per step, from the library's own count of timed scans (the bf16 prefilter
        # takes up to 2048 queries a pass in its query-stationary form and 256 in its tiled form)
        # (one query a step is one pass whatever was timed: a handle with deferred selects reports every launch,
        # timed in runs, not every prof_every-th)
        passes = 1 if (args.hybrid or B == 1) else max(1, round(scan_launches * prof_every / max(args.steps, 1)))
        b_launch = B / passes                                   # queries served by one scan launch
        flops = 2.0 * b_launch * n_local * d
        tflops = flops / (scan_avg_ms * 1e-3) / 1e12 if scan_avg_ms > 0 else 0.0
        # which roofline binds: 2*B*N*d flops over N*d*4 bytes against the peak of the matrix
        # pipe the path uses (f32 157.3 TF: B >= 40; bf16 prefilter 2500 TF: B >= 625) and 8 TB/s
        bf16 = mode_used == 4
        mfma_peak = 2500.0 if bf16 else 157.3
        mfma_bound = mode_used in (3, 4) and b_launch / 2.0 >= mfma_peak * 1e12 / (HBM_PEAK_GBS * 1e9)
        result = {
            "metric": f"queries/sec, brute-force {args.metric} top-10 over N x d fp32 (achieved HBM GB/s in roofline)",
            "value": args.steps * B / elapsed,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU over gloo, not a measurement)" if one_gpu else ""),
            "config": {"workload": (f"{n_total}x{d} fp32 corpus in HBM, hybrid step: dense cosine top-50 + BM25 "
                                    f"top-50 over {n_total} synthetic documents (both on the GPU) -> RRF(60) on "
                                    f"the device -> top-{args.k}; no host round trip in the step") if args.hybrid else
                       f"{n_total}x{d} fp32 corpus in HBM, {B} quer{'y' if B == 1 else 'ies'} per step, {args.metric} top-{k}",
                       "rows": n_total, "dim": d, "k": k, "batch": B, "metric": args.metric,
                       "rows_per_gpu": n_local, "parallelism": f"row-shard x{world}",
                       "exchange": ("one trag_shard_search_device call per step: local scan, records, RCCL "
                                    "all-gather (version %d), merge on every rank; exchange of step i overlaps "
                                    "the scan of step i+1" % rccl_version_or_zero()) if group is not None else
                                   ("Python ShardExchange over gloo (rehearsal)" if ex is not None else None),
                       "rccl_version": rccl_version_or_zero() if group is not None else None,
                       "rccl_comm_world": (ranks_info[0].get("comm_world") if ranks_info else None),
                       "ranks": ranks_info,
                       "scan_mode": {0: "auto", 1: "exact", 2: "stream", 3: "mfma", 4: "bf16-prefilter", 5: "bf16-mirror"}[args.mode],
                       "bf16_mirror": bool(args.mirror),
                       "entry": "host pointers (PCIe + synchronisation inside every step)" if args.host
                                else "device-resident queries and results",
                       "pipelined": (("select of query i as one more workgroup of the launch of query i+1's scan; "
                                      if (pipe_mode == 2 or group is not None) else
                                      "select of query i on the handle's side stream beside the scan of query i+1; ")
                                     + "all results fenced inside the timed region") if pipelined else None,
                       "latency_us_per_call": 1e3 * ms_per_step if args.host else None,
                       "path_used": {1: "exact scan (f32, reference order)", 2: "streaming scan + exact re-score",
                                     3: "f32 matrix cores + exact re-score",
                                     4: "bf16 matrix-core prefilter + exact f32 scoring of candidates",
                                     5: "scan of the bf16 mirror + exact f32 scoring of the rows within its bound"
                                     }.get(mode_used, str(mode_used))},
            "roofline": ({"bound": "mfma", "achieved": tflops, "peak": mfma_peak, "unit": "TFLOP/s",
                          "frac": tflops / mfma_peak, "traffic": None, "flops_per_launch": flops}
                         if mfma_bound else
                         {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": achieved / HBM_PEAK_GBS, "traffic": None}),

            "parity": parity,
        }
        if sparse_info is not None:
            t0 = time.perf_counter()
            for i in range(20):
                bm25.search_ids(q_terms[i % len(q_terms)], cand)
            sparse_info["search_ms_alone"] = 1e3 * (time.perf_counter() - t0) / 20
            result["config"]["sparse"] = sparse_info
        if batch_leg is not None:
            result["batch_1000"] = batch_leg
        if traffic is not None:
            result["roofline"]["traffic"] = traffic.pop("bytes_per_launch")
            result["roofline"]["traffic_source"] = traffic
        if world == 1 and not args.hybrid:
            probe = store.stream_probe(3)   # plain contiguous read of the same rows, on this box, right now
            if probe > 0:
                result["roofline"]["measured_stream_read_GBps"] = probe
                if result["roofline"]["bound"] == "hbm":
                    result["roofline"]["frac_of_measured_stream_read"] = result["roofline"]["achieved"] / probe
        result["roofline"].update({"kernel": "scan (corpus pass)", "kernel_avg_ms": scan_avg_ms,
                                   "launches_timed": scan_launches, "queries_per_launch": b_launch,
                                   "algorithmic_bytes_per_launch": algo_bytes,
                                   "whole_step_GBps": algo_bytes / (ms_per_step * 1e-3) / 1e9,
                                   # a kernel cannot take longer than the step that contains it: if the event
                                   # brackets say so, they measured something else and `frac` is not evidence
                                   "timing_inconsistent": bool(scan_avg_ms * passes > ms_per_step * 1.005)})
        if pipe_mode == 2 and group is None and ex is None:
            result["roofline"]["kernel_timed"] = ("event pairs around runs of 4 consecutive launches (a launch = this query's "
                                                  "scan + the previous query's select as one more workgroup): a pair around "
                                                  "a single launch opens a gap that the steps do not have")
        if args.hybrid:
            result["roofline"]["kernel_timed"] = ("the dense scan alone on the stream (no BM25 beside it), the last "
                                                  "steps' queries again after the timed region")
        if world == 1 and not args.hybrid and not args.mirror and args.mode == 0 and not args.host:
            result["with_bf16_mirror"] = mirror_leg(store, args, Q, out_rows, out_scores, out_n, status, n_local, d,
                                                    B, k, esz, sptr)
        if not args.no_cpu_baseline and world == 1:   # the CPU leg belongs to the single-GPU line only
            result["cpu_baseline"] = cpu_baseline(store, Q, n_local, n_total, d, k, args.cpu_sample_rows,
                                                  seed, row0, metric_id)
        os.write(result_fd, (json.dumps(result) + "\n").encode())
    if group is not None:
        group.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def rccl_version_or_zero():
    try:
        from trueno_rag_amd.sharded import rccl_version
        return rccl_version()
    except Exception:  # noqa: BLE001
        return 0


def exchange_batch_leg(args, store, group, ex, side, rank, world, row0, d, k, seed, sptr, stream, fence, dist):
    """Second leg of a sharded run: 1000 queries per step (the batch of the 8-GPU benchmark configuration) through the
    same local search -> all-gather -> merge step. Reported beside the headline leg."""
    BX = 1000
    steps, warm = max(2, min(args.steps, 10)), 1
    n_q = (steps + warm) * BX
    Q = torch.from_numpy(unit_queries(n_q, d, seed + 200)).cuda()
    rows = torch.zeros((n_q, k), dtype=torch.int32, device="cuda")
    scores = torch.zeros((n_q, k), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(n_q, dtype=torch.int32, device="cuda")
    status = torch.zeros(n_q, dtype=torch.int32, device="cuda")
    if ex is not None:
        l_rows, l_scores, l_cnt = torch.zeros_like(rows), torch.zeros_like(scores), torch.zeros_like(cnt)

    def step(s):
        i = s * BX
        if group is not None:
            group.search_device(store, row0, 1, Q.data_ptr() + i * d * 4, BX, k, rows.data_ptr() + i * k * 4,
                                scores.data_ptr() + i * k * 4, cnt.data_ptr() + i * 4, status.data_ptr() + i * 4, sptr)
            return
        store.search_device(Q.data_ptr() + i * d * 4, BX, k, l_rows.data_ptr() + i * k * 4,
                            l_scores.data_ptr() + i * k * 4, l_cnt.data_ptr() + i * 4, status.data_ptr() + i * 4, sptr)
        done = torch.cuda.Event()
        done.record(stream)
        side.wait_event(done)
        with torch.cuda.stream(side):
            ex.exchange(l_rows[i:i + BX], l_scores[i:i + BX], l_cnt[i:i + BX], row0, rows[i:i + BX], scores[i:i + BX],
                        cnt[i:i + BX], side.cuda_stream)

    for s in range(warm):
        step(s)
    fence()
    t0 = time.perf_counter()
    for s in range(warm, warm + steps):
        step(s)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    unfinished = int((status[warm * BX:] != 0).sum().item())
    # every rank must hold the same merged lists: compare a checksum of this rank's with rank 0's
    digest = [int(rows[warm * BX:].to(torch.int64).sum().item()), int(cnt.sum().item()),
              float(scores[warm * BX:].double().sum().item())]
    same = True
    if world > 1:
        all_d = [None] * world
        dist.all_gather_object(all_d, digest)
        same = all(x == all_d[0] for x in all_d)
    sc = scores[-1].cpu().numpy()
    # the first 32 queries of the last step with their merged lists, for the single-store check
    lo = (warm + steps - 1) * BX
    sample = (Q[lo:lo + 32].cpu().numpy(), rows[lo:lo + 32].cpu().numpy().astype(np.uint32),
              scores[lo:lo + 32].cpu().numpy(), cnt[lo:lo + 32].cpu().numpy(), status[lo:lo + 32].cpu().numpy())
    return ({"value": steps * BX / elapsed, "unit": "queries/s", "batch": BX, "steps": steps,
             "ms_per_step": 1e3 * elapsed / steps, "unfinished_queries": unfinished,
             "sorted": bool(np.all(sc[:-1] >= sc[1:])), "all_ranks_hold_the_same_lists": bool(same),
             "path_used": store.last_stats()["mode_used"]}, sample)


def mirror_leg(store, args, Q, out_rows, out_scores, out_n, status, n_local, d, B, k, esz, sptr):
    """The same steps once more with the optional bf16 copy of the corpus in HBM (1.5 x the
    memory): passes read the copy, the rows that can be in the top k under the rounding bound
    are scored from the f32 corpus, and the results must equal the headline run's bit for bit.
    Reported beside the headline number, never instead of it."""
    free, _ = torch.cuda.mem_get_info()
    if free < n_local * d * 2 * 1.2:
        return {"skipped": "not enough free HBM for the bf16 copy"}
    ref_rows, ref_scores = out_rows.clone(), out_scores.clone()
    store.set_bf16_mirror(True)
    out_rows.zero_(); out_scores.zero_(); status.fill_(9)

    def step(s):
        i = s * B
        store.search_device(Q.data_ptr() + i * d * esz, B, k, out_rows.data_ptr() + i * k * esz,
                            out_scores.data_ptr() + i * k * esz, out_n.data_ptr() + i * esz,
                            status.data_ptr() + i * esz, sptr)

    for i in range(args.warmup):
        step(i)
    store.fence(sptr)
    torch.cuda.synchronize()
    store.set_profiling(4 if args.steps >= 40 else 1)
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        step(i)
    store.fence(sptr)          # (a pipelined handle: the stages behind the last scans)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    scan_ms, launches = store.scan_time()
    store.set_profiling(0)
    mode_used = store.last_stats()["mode_used"]
    store.set_bf16_mirror(False)
    lo = args.warmup * B
    same = bool(torch.equal(out_rows[lo:], ref_rows[lo:]) and
                torch.equal(out_scores[lo:].view(torch.int32), ref_scores[lo:].view(torch.int32)))
    avg = scan_ms / max(launches, 1)
    bytes_launch = float(n_local) * d * 2.0
    return {"value": args.steps * B / elapsed, "unit": "queries/s", "ms_per_step": 1e3 * elapsed / args.steps,
            "scan_kernel_avg_ms": avg, "bytes_per_launch": bytes_launch,
            "scan_GBps": bytes_launch / (avg * 1e-3) / 1e9 if avg > 0 else 0.0,
            "frac_of_hbm_peak": bytes_launch / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS if avg > 0 else 0.0,
            "path_used": {4: "bf16 matrix-core prefilter on the mirror + exact f32 scoring",
                          5: "scan of the bf16 mirror + exact f32 scoring"}.get(mode_used, str(mode_used)),
            "unfinished_queries": int((status[lo:] != 0).sum().item()),
            "identical_to_headline_results": same, "extra_hbm_bytes": bytes_launch}


def pmc_traffic(args):
    """FETCH_SIZE of the scan kernel, per launch, from a counters-only rocprofv3 pass over a
    3-step copy of this run (its own process). FETCH_SIZE counts KiB at the L2's memory side;
    on gfx950 it tallies the 128-byte requests of a coalesced 16-byte-per-lane stream at 64
    bytes, so the byte count is 2 * 1024 * FETCH_SIZE. Returns None when the pass is not
    possible (no rocprofv3, a failure, a timeout): the bench line then carries traffic null."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    out = tempfile.mkdtemp(prefix="trag_pmc_", dir="/tmp")
    cmd = [exe, "--pmc", "FETCH_SIZE", "--output-format", "csv", "-d", out, "-o", "p", "--",
           sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
           "--no-pmc", "--rows", str(args.rows), "--dim", str(args.dim), "--k", str(args.k),
           "--mode", str(args.mode), "--batch", str(args.batch), "--metric", args.metric] + (
               ["--mirror"] if args.mirror else [])
    env = dict(os.environ, TRAG_BENCH_CHILD="1", TMPDIR="/tmp")
    try:
        proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                start_new_session=True)
        try:
            proc.wait(timeout=150)
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)
            proc.wait()
            return None
        vals = {}
        for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                name = r.get("Kernel_Name", "")
                if r.get("Counter_Name") != "FETCH_SIZE":
                    continue
                if ("scan_exact" in name or "scan_mirror" in name or "scan_stream" in name or "gemm_" in name):
                    key = name.split("(")[0]
                    if "scan_exact_v4" in key:   # (with and without the previous query's select aboard: one kernel)
                        key = key.replace(", true>", ">").replace(", false>", ">")
                    vals.setdefault(key, []).append(float(r["Counter_Value"]))
        if not vals:
            return None
        # the dominant scan kernel = the one that fetched most in total; drop its first launch (warm-up)
        name, v = max(vals.items(), key=lambda kv: sum(kv[1]))
        v = v[1:] if len(v) > 1 else v
        kib = sum(v) / len(v)
        return {"bytes_per_launch": 2.0 * 1024.0 * kib, "counter": "FETCH_SIZE (KiB) x 1024 x 2 (gfx950 tallies "
                "128-byte requests at 64 bytes)", "kernel": name.replace("void ", ""), "launches": len(v),
                "tool": "rocprofv3 --pmc FETCH_SIZE, separate 3-step run of the same workload"}
    except Exception:  # noqa: BLE001 -- the counter pass is optional evidence, never a reason to fail the bench
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def sparse_queries(n, vocab, n_terms, seed):
    """n queries of n_terms term ids, Zipf(1.0) over the vocabulary (numpy, seeded)."""
    rng = np.random.default_rng(seed)
    cdf = np.cumsum(1.0 / np.arange(1, vocab + 1))
    u = rng.random((n, n_terms)) * cdf[-1]
    return np.minimum(np.searchsorted(cdf, u, side="right"), vocab - 1).astype(np.uint32)


def cpu_baseline(store, Q, n_local, n_total, d, k, sample_rows, seed, first_global_row, metric_id=0):
    """The reference's CPU path (restated in oracle/, single thread as the
    reference is) timed on a bounded slice of the same corpus: about 10 s of CPU work."""
    from oracle import oracle
    m = min(sample_rows, n_local)
    sample = store.read_rows(0, m)
    n_q = 16
    qs = Q[-n_q:].cpu().numpy()
    sec, _, _ = oracle.bench_faithful(sample, qs, k, metric_id)
    per_query_full = sec * (n_total / m)
    # not the reference, for scale only: the same arithmetic over one contiguous N x d array
    # with the row loop spread over the host cores (OpenMP) and a partial selection instead of
    # the full sort
    t = []
    for i in range(3):
        t0 = time.perf_counter()
        oracle.search(sample, qs[i], k, metric_id)
        t.append(time.perf_counter() - t0)
    fast = sorted(t)[1] * (n_total / m)
    return {"value": 1.0 / per_query_full, "unit": "queries/s", "cores": 1, "kind": "port",
            "sample": f"first {m} rows of the corpus, {n_q} queries, median seconds/query {sec:.4f}, "
                      f"scaled linearly to {n_total} rows",
            "host_cpus_visible": os.cpu_count(),
            "parity_vs_oracle": oracle_parity(sample, qs, d, k, seed, first_global_row, metric_id,
                                              store.config().device),
            "not_the_reference": {"what": "contiguous layout, row loop over all host threads (OpenMP), partial "
                                          "selection; same f32 arithmetic",
                                  "value": 1.0 / fast, "unit": "queries/s", "threads": oracle.num_threads()}}


def oracle_parity(sample, qs, d, k, seed, first_global_row, metric_id, device):
    """The library against the CPU oracle at the sample's size: a second store that holds the same m rows
    (the generator is a pure function of (seed, row)), searched through the automatic routing."""
    from oracle import oracle
    pkg = load_package()
    m, n_q = sample.shape[0], qs.shape[0]
    sub = pkg.VectorStore(pkg.VectorStoreConfig(dimension=d, metric=pkg.DistanceMetric(metric_id),
                                                device=device, reserve_rows=m))
    sub.fill_synthetic(m, seed, first_global_row=first_global_row)
    same_rows, same_bits, max_diff, ties = True, True, 0.0, 0
    for i in range(n_q):
        er, es = oracle.search(sample, qs[i], k, metric_id)
        gr, gs = sub.search_rows(qs[i], k)
        same_rows = same_rows and np.array_equal(gr, er)
        same_bits = same_bits and gs.tobytes() == es.tobytes()
        max_diff = max(max_diff, float(np.max(np.abs(gs - es))) if len(es) == len(gs) and len(es) else 0.0)
        ties += int(np.sum(es[1:] == es[:-1]))
    sub.close()
    return {"rows": m, "queries": n_q, "k": k, "ranks_identical": bool(same_rows), "score_bits_identical": bool(same_bits),
            "max_abs_score_diff": max_diff, "exact_score_ties_seen": ties}


def single_store_check(pkg, device, n_total, d, k, seed, metric_id, head, batch_sample, oracle_rows, with_oracle=True):
    """Rank 0 of a sharded run: the whole corpus once more in ONE store on this GPU, scanned in the
    reference's order (scan mode `exact`), and the sharded run's merged lists compared with its
    results bit for bit -- rows, score bits and counts. `head`: (queries, rows, scores, counts,
    statuses) of the headline leg's last queries; `batch_sample`: the same of the 1000-query leg."""
    out = {}
    need = n_total * d * 4 * 1.12 + n_total * 16 + (1 << 28)
    free, _ = torch.cuda.mem_get_info()
    if free < need:
        out["sharded_equals_single_store"] = None
        out["single_store_check_skipped"] = f"{free / 1e9:.1f} GB free on rank 0's GPU, the whole corpus needs {need / 1e9:.1f}"
        return out
    t0 = time.perf_counter()
    full = pkg.VectorStore(pkg.VectorStoreConfig(dimension=d, metric=pkg.DistanceMetric(metric_id), device=device,
                                                 scan_mode=1, reserve_rows=n_total))
    full.fill_synthetic(n_total, seed, first_global_row=0)

    def same(sample):
        qs, rows, scores, counts, statuses = sample
        fr, fs, fn = full.search_batch_rows(qs, k)
        ok, checked, unfinished = True, 0, 0
        for b in range(qs.shape[0]):
            if statuses[b] != 0:       # a query the device entry left unfinished is counted, not compared
                unfinished += 1
                continue
            n = int(fn[b])
            ok = ok and int(counts[b]) == n and np.array_equal(rows[b, :n], fr[b, :n]) \
                and scores[b, :n].tobytes() == fs[b, :n].tobytes()
            checked += 1
        return ok, checked, unfinished

    ok_h, n_h, u_h = same(head)
    detail = {"whole_corpus_rows": n_total, "route": "exact-order scan of one store on rank 0's GPU",
              "headline_queries_compared": n_h, "headline_identical": bool(ok_h), "unfinished_skipped": u_h}
    ok = ok_h and n_h > 0
    if batch_sample is not None:
        ok_b, n_b, u_b = same(batch_sample)
        detail.update({"batch_1000_queries_compared": n_b, "batch_1000_identical": bool(ok_b)})
        detail["unfinished_skipped"] += u_b
        ok = ok and ok_b and n_b > 0
    out["sharded_equals_single_store"] = bool(ok)
    if with_oracle:
        m = min(oracle_rows, n_total)
        out["parity_vs_oracle"] = oracle_parity(full.read_rows(0, m), head[0], d, k, seed, 0, metric_id, device)
    full.close()
    detail["seconds"] = time.perf_counter() - t0
    out["single_store_check"] = detail
    return out


if __name__ == "__main__":
    main()
