"""One-off check at the bench's full size: for many random queries over the 10M x 384 synthetic corpus the
exact scan (sparse key stores on), the scan of the bf16 mirror and the streaming scan (every key stored) must
return the same rows and score bits. Usage: python scripts/hunt_full_size.py [queries]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
n, d = 10_000_000, 384
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 40
store = pkg.VectorStore(pkg.VectorStoreConfig(dimension=d, reserve_rows=n))
store.fill_synthetic(n, 1004)
store.set_bf16_mirror(True)
rng = np.random.default_rng(7)
bad = 0
for i in range(nq):
    q = rng.standard_normal(d).astype(np.float32)
    if i % 4 == 0:   # close to a stored row: a clear winner and its neighbours
        q = store.read_rows(int(rng.integers(n)), 1)[0] + np.float32(0.02) * q
    k = int(rng.choice([1, 10, 50, 64]))
    out = {}
    for mode in (1, 5, 2):
        store.set_scan_mode(mode)
        out[mode] = store.search_rows(q, k)
        if mode == 1:
            kept = store.last_stats()["keys_kept_from"]
    for mode in (5, 2):
        same = np.array_equal(out[1][0], out[mode][0]) and out[1][1].tobytes() == out[mode][1].tobytes()
        if not same:
            bad += 1
            print(f"query {i} k {k}: mode {mode} differs from the exact scan", flush=True)
    if i % 10 == 0:
        print(f"query {i} k {k} kept_from {kept:#x} ok so far: {bad == 0}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
