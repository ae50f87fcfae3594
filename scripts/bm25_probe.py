"""GPU-box probe: time every synthetic BM25 query of the hybrid bench and print the slow ones."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
from bench import sparse_queries
pkg = load_package()
bm = pkg.BM25Index(device=0)
bm.fill_synthetic(2_000_000, 50000, 64, 1005, first_global_doc=0)
bm.search_ids([0], 1)
q = sparse_queries(55, 50000, 8, 1105)
for i in range(55):
    t0 = time.perf_counter()
    d, s = bm.search_ids(q[i], 50)
    dt = 1e3 * (time.perf_counter() - t0)
    if dt > 1.0 or i < 3:
        print(i, round(dt, 2), q[i].tolist(), len(d))
