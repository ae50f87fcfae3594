#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_search_fixture.npz from the CPU oracle.

The reference is Rust and cannot run here, so these vectors pin the oracle (and
through it the HIP path) against drift; the reference's own known answers are
in ref_unit_vectors.json / ref_fusion_vectors.json.  Usage: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as o  # noqa: E402

out = {}
for d, n, seed in ((3, 64, 11), (128, 512, 12), (384, 1024, 13)):
    corpus = o.synth_rows(seed, 0, n, d)
    queries = o.synth_rows(seed + 100, 0, 4, d)
    out[f"seed_{d}"] = np.int64(seed)
    out[f"n_{d}"] = np.int64(n)
    for m in range(3):
        rr, ss = [], []
        for qi in range(4):
            r, s = o.search(corpus, queries[qi], 10, m)
            rr.append(r)
            ss.append(s)
        out[f"rows_{d}_{m}"] = np.stack(rr)
        out[f"scores_{d}_{m}"] = np.stack(ss)
np.savez_compressed(os.path.join(HERE, "oracle_search_fixture.npz"), **out)
print("wrote oracle_search_fixture.npz")
