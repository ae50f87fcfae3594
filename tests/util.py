"""Shared checks for the parity tests."""
import numpy as np


def assert_same_ranking(got_rows, got_scores, exp_rows, exp_scores, what=""):
    """Bit-exact parity: identical row order and identical score bits (the two
    zeros compare equal under the reference's partial_cmp and are treated so)."""
    got_rows = np.asarray(got_rows, dtype=np.uint32)
    exp_rows = np.asarray(exp_rows, dtype=np.uint32)
    got_scores = np.asarray(got_scores, dtype=np.float32)
    exp_scores = np.asarray(exp_scores, dtype=np.float32)
    assert got_rows.shape == exp_rows.shape, f"{what}: {got_rows.shape} results, expected {exp_rows.shape}"
    if not np.array_equal(got_rows, exp_rows):
        bad = np.nonzero(got_rows != exp_rows)[0][:5]
        raise AssertionError(f"{what}: rank mismatch at positions {bad.tolist()}: got rows "
                             f"{got_rows[bad].tolist()} scores {got_scores[bad].tolist()}, expected rows "
                             f"{exp_rows[bad].tolist()} scores {exp_scores[bad].tolist()}")
    g, e = got_scores.copy(), exp_scores.copy()
    g[g == 0] = 0.0  # -0.0 -> +0.0
    e[e == 0] = 0.0
    same = (g.view(np.uint32) == e.view(np.uint32)) | (np.isnan(g) & np.isnan(e))
    if not same.all():
        bad = np.nonzero(~same)[0][:5]
        raise AssertionError(f"{what}: score bits differ at {bad.tolist()}: got {got_scores[bad].tolist()} "
                             f"expected {exp_scores[bad].tolist()}")


def unit_rows(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True).astype(np.float32)
    return np.ascontiguousarray(x, dtype=np.float32)
