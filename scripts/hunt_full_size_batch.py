"""One-off check at the bench's full size: batches of many sizes through the automatic routing (multi-query exact pass,
bf16 prefilter in its tiled and query-stationary forms, with and without the bf16 mirror) over the 10M x 384 synthetic
corpus must return, query by query, the rows and score bits of the exact single-query scan.
Usage: python scripts/hunt_full_size_batch.py [rounds]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
n, d = 10_000_000, 384
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
store = pkg.VectorStore(pkg.VectorStoreConfig(dimension=d, reserve_rows=n))
store.fill_synthetic(n, 1004)
rng = np.random.default_rng(11)
bad = 0
for rnd in range(rounds):
    for mirror in (False, True):
        store.set_bf16_mirror(mirror)
        for B in (2, 3, 5, 8, 17, 64, 130, 300):
            k = int(rng.choice([1, 10, 50, 100]))
            Q = rng.standard_normal((B, d)).astype(np.float32)
            for j in range(0, B, 4):   # close to a stored row: a clear winner and its neighbours
                Q[j] = store.read_rows(int(rng.integers(n)), 1)[0] + np.float32(0.02) * Q[j]
            store.set_scan_mode(0)
            rows, scores, cnt = store.search_batch_rows(Q, k)
            used = store.last_stats()["mode_used"]
            store.set_scan_mode(1)
            picks = sorted({0, 1, B // 2, B - 1} | {int(x) for x in rng.integers(0, B, 4)})
            for b in picks:
                er, es = store.search_rows(Q[b], k)
                if not (np.array_equal(rows[b][:cnt[b]], er) and scores[b][:cnt[b]].tobytes() == es.tobytes()):
                    bad += 1
                    print(f"round {rnd} mirror {mirror} B {B} k {k} route {used}: query {b} differs from the exact scan", flush=True)
            print(f"round {rnd} mirror {mirror} B {B} k {k} route {used}: ok so far {bad == 0}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
