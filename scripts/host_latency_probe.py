"""GPU-box probe: latency of the host-pointer entry point (trag_store_search) on small stores --
the smallest benchmark configuration (10 000 x 384, one query, top 10) and two larger sizes."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
for n in (10_000, 100_000, 1_000_000):
    st = pkg.VectorStore(pkg.VectorStoreConfig(dimension=384, reserve_rows=n))
    st.fill_synthetic(n, 1001)
    rng = np.random.default_rng(1)
    Q = rng.standard_normal((64, 384)).astype(np.float32)
    for i in range(20):
        st.search_rows(Q[i % 64], 10)
    t = []
    for i in range(500):
        t0 = time.perf_counter()
        st.search_rows(Q[i % 64], 10)
        t.append(time.perf_counter() - t0)
    t.sort()
    print(f"{n:>8} rows: median {1e6 * t[len(t) // 2]:.1f} us, p10 {1e6 * t[len(t) // 10]:.1f} us, p90 {1e6 * t[9 * len(t) // 10]:.1f} us")
    st.close()
