#!/usr/bin/env python3
"""Writes tests/golden/config_vectors.npz: expected top-k rows and score bits for seeded
corpora at the dimensions of the benchmark configurations, adversarial small cases and
reciprocal-rank-fusion cases.

Everything here is computed with numpy ONLY -- a second, independent restatement of the
reference's arithmetic (src/index.rs:386-462, src/fusion.rs:68-84, 227-231 of the reference):
it shares no code with oracle/trag_oracle.c or with the HIP kernels, so agreement of all three,
bit for bit, is a cross-check and not a tautology. Rust's f32 iterator sums are sequential left
folds with the product and the addition rounded separately; numpy float32 arithmetic on arrays
rounds every elementwise operation once, so looping over the columns (vectorised over the
rows) reproduces that order exactly.

    python tests/golden/make_golden_configs.py
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
f32 = np.float32
M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


# ---- the synthetic corpus generator (same integer hash as the device fill kernel) ----
def splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & M64
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M64
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M64
    return x ^ (x >> np.uint64(31))


def synth_rows(seed, row0, n, d):
    with np.errstate(over="ignore"):
        rows = (np.arange(n, dtype=np.uint64) + np.uint64(row0))[:, None]
        cols = np.arange(d, dtype=np.uint64)[None, :]
        h = splitmix64(np.uint64(seed) ^ splitmix64(rows * np.uint64(0x100000001B3) + cols))
    s = ((h & np.uint64(0xFFFF)).astype(np.int64) + ((h >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.int64) +
         ((h >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64) +
         ((h >> np.uint64(48)) & np.uint64(0xFFFF)).astype(np.int64) - 131070)
    v = s.astype(f32) * f32(1.0 / 37837.0)
    ss = np.zeros(n, dtype=f32)
    for c in range(d):                      # sequential sum of squares, one rounding per operation
        ss = ss + v[:, c] * v[:, c]
    nrm = np.sqrt(ss)
    safe = np.where(nrm > 0, nrm, f32(1.0))
    return np.where((nrm > 0)[:, None], v / safe[:, None], v).astype(f32)


# ---- the three metrics, reference order ----
def scores(corpus, q, metric):
    n, d = corpus.shape
    with np.errstate(all="ignore"):
        if metric == 1:                     # euclidean, negated (src/index.rs:400,452-458)
            acc = np.zeros(n, dtype=f32)
            for j in range(d):
                t = q[j] - corpus[:, j]
                acc = acc + t * t
            return -np.sqrt(acc)
        dot = np.zeros(n, dtype=f32)
        nb = np.zeros(n, dtype=f32)
        na = f32(0.0)
        for j in range(d):
            dot = dot + q[j] * corpus[:, j]
            if metric == 0:
                nb = nb + corpus[:, j] * corpus[:, j]
                na = f32(na + f32(q[j] * q[j]))
        if metric == 2:                     # dot product (src/index.rs:460-462)
            return dot
        na, nb = np.sqrt(na), np.sqrt(nb)   # cosine: dot / (na * nb), zero norm -> 0 (src/index.rs:440-450)
        den = (na * nb).astype(f32)
        out = np.where((na == 0) | (nb == 0), f32(0.0), dot / np.where(den == 0, f32(1.0), den))
        return out.astype(f32)


def top_k(sc, k, dead=None):
    """Best first; equal scores by ascending row; NaN after every number; removed rows never."""
    idx = np.arange(sc.shape[0])
    if dead is not None:
        idx = idx[~dead.astype(bool)]
    s = sc[idx]
    nan = np.isnan(s)
    key = np.where(nan, -np.inf, s).astype(np.float64)
    order = np.lexsort((idx, -key, nan))    # primary: numbers before NaN; then score desc; then row asc
    order = order[:k]
    return idx[order].astype(np.uint32), s[order].astype(f32)


def rrf(dense_ids, sparse_ids, k):
    """src/fusion.rs:68-84 + 227-231: score += 1/((k + rank) + 1) in f32, dense list first;
    descending, equal fused scores in order of first appearance."""
    table, order = {}, []
    for ids in (dense_ids, sparse_ids):
        for rank, i in enumerate(ids):
            inc = f32(1.0) / f32(f32(f32(k) + f32(rank)) + f32(1.0))
            if i not in table:
                table[i] = f32(0.0)
                order.append(i)
            table[i] = f32(table[i] + inc)
    pos = {i: p for p, i in enumerate(order)}
    out = sorted(order, key=lambda i: (-float(table[i]), pos[i]))
    return np.array(out, dtype=np.uint64), np.array([table[i] for i in out], dtype=f32)


def main():
    out = {}
    # (ii) the benchmark dimensions: N = 4096 seeded rows, 8 queries, top 100 (k = 1, 10, 50 are prefixes)
    for d, seed in ((384, 2001), (768, 2002), (4096, 2003)):
        corpus = synth_rows(seed, 0, 4096, d)
        queries = synth_rows(seed + 100, 0, 8, d)
        out[f"cfg_{d}_seed"] = np.int64(seed)
        out[f"cfg_{d}_corpus_head"] = corpus[:4].copy()      # pins the generator itself
        for m in range(3):
            rr, ss = zip(*(top_k(scores(corpus, queries[b], m), 100) for b in range(8)))
            out[f"cfg_{d}_{m}_rows"] = np.stack(rr)
            out[f"cfg_{d}_{m}_scores"] = np.stack(ss)
    # (iii) adversarial: duplicates, zero rows, zero query, NaN row, removed rows, k > N, k = 0
    rng = np.random.default_rng(7)
    adv = rng.standard_normal((40, 8)).astype(f32)
    adv[5] = adv[3]                # duplicate
    adv[11] = adv[3]               # duplicate again
    adv[7] = 0.0                   # zero-norm row
    adv[13, 2] = np.nan            # NaN row
    adv[17] = adv[3] * f32(2.0)    # same cosine, different dot / distance
    dead = np.zeros(40, dtype=np.uint8)
    dead[[9, 21]] = 1
    aq = np.stack([adv[3], np.zeros(8, dtype=f32), rng.standard_normal(8).astype(f32)])
    out["adv_corpus"], out["adv_dead"], out["adv_queries"] = adv, dead, aq
    for m in range(3):
        for b in range(3):
            for k in (0, 5, 40, 64):
                r, s = top_k(scores(adv, aq[b], m), k, dead)
                out[f"adv_{m}_{b}_{k}_rows"], out[f"adv_{m}_{b}_{k}_scores"] = r, s
    # (iv) RRF: two 50-lists, 15 shared ids, including ids that tie exactly (rank i in one list only)
    dense = np.arange(100, 150, dtype=np.uint64)
    sparse = np.concatenate([np.arange(135, 150, dtype=np.uint64)[::-1], np.arange(300, 335, dtype=np.uint64)])
    out["rrf_dense"], out["rrf_sparse"] = dense, sparse
    for k in (60.0, 1.0):
        ids, sc = rrf(dense.tolist(), sparse.tolist(), k)
        out[f"rrf_{int(k)}_ids"], out[f"rrf_{int(k)}_scores"] = ids, sc
    np.savez_compressed(os.path.join(HERE, "config_vectors.npz"), **out)
    print("wrote config_vectors.npz,", len(out), "arrays")


if __name__ == "__main__":
    main()
