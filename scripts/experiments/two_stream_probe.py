"""Timing probe: single-query device-entry searches on ONE stream (pipelined modes 2 and 1) against the same calls
alternating between TWO caller streams in mode 1 (two scans in flight: the tail of one beside the head of the next).
usage: python scripts/experiments/two_stream_probe.py [rows] [dim]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 384
k, N = 10, 400
store = pkg.VectorStore(pkg.VectorStoreConfig(dimension=d, device=0, reserve_rows=n))
store.fill_synthetic(n, 1004)
rng = np.random.default_rng(3)
Q = rng.standard_normal((N, d)).astype(np.float32)
dq = torch.from_numpy(Q).cuda()
rows = torch.zeros((N, k), dtype=torch.int32, device="cuda")
scores = torch.zeros((N, k), dtype=torch.float32, device="cuda")
cnt = torch.zeros(N, dtype=torch.int32, device="cuda")
status = torch.zeros(N, dtype=torch.int32, device="cuda")
A, B = torch.cuda.Stream(), torch.cuda.Stream()


def run(mode, two):
    store.set_pipelined(mode)
    best = 1e9
    for rep in range(4):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(N):
            st = (B if (two and i & 1) else A).cuda_stream
            store.search_device(dq.data_ptr() + i * d * 4, 1, k, rows.data_ptr() + i * k * 4, scores.data_ptr() + i * k * 4,
                                cnt.data_ptr() + i * 4, status.data_ptr() + i * 4, st)
        store.fence(A.cuda_stream)
        store.fence(B.cuda_stream)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / N * 1e3)
    return best, rows.clone()


ref = None
for name, mode, two in (("mode 2, one stream", 2, False), ("mode 1, one stream", 1, False), ("mode 1, two streams", 1, True),
                        ("mode 0, two streams", 0, True)):
    ms, r = run(mode, two)
    if ref is None:
        ref = r
    print(f"{n} x {d}  {name}: {ms:.4f} ms a query, {n * d * 4 / ms / 1e6:.0f} GB/s, same rows as the first: {bool(torch.equal(r, ref))}", flush=True)
