"""tests/golden/config_vectors.npz holds expected results computed by a numpy-only restatement
of the reference's arithmetic (tests/golden/make_golden_configs.py), independent of the C oracle
and of the HIP kernels. CPU: the C oracle and the host fusion code must reproduce it bit for bit.
GPU: so must every scan mode of the HIP path, on a corpus the device generates itself."""
import os

import numpy as np
import pytest

from util import assert_same_ranking

HERE = os.path.dirname(os.path.abspath(__file__))
FX = np.load(os.path.join(HERE, "golden", "config_vectors.npz"))
DIMS = (384, 768, 4096)


@pytest.mark.parametrize("d", DIMS)
def test_oracle_reproduces_config_vectors(oracle, d):
    seed = int(FX[f"cfg_{d}_seed"])
    corpus = oracle.synth_rows(seed, 0, 4096, d)
    assert corpus[:4].tobytes() == FX[f"cfg_{d}_corpus_head"].tobytes()
    Q = oracle.synth_rows(seed + 100, 0, 8, d)
    for m in range(3):
        for b in range(8):
            for k in (1, 10, 50, 100):
                rows, scores = oracle.search(corpus, Q[b], k, m)
                assert_same_ranking(rows, scores, FX[f"cfg_{d}_{m}_rows"][b][:k], FX[f"cfg_{d}_{m}_scores"][b][:k],
                                    f"oracle d={d} metric={m} q={b} k={k}")


def test_oracle_reproduces_adversarial_vectors(oracle):
    corpus, dead, Q = FX["adv_corpus"], FX["adv_dead"], FX["adv_queries"]
    for m in range(3):
        for b in range(3):
            for k in (0, 5, 40, 64):
                rows, scores = oracle.search(corpus, Q[b], k, m, dead=dead)
                assert_same_ranking(rows, scores, FX[f"adv_{m}_{b}_{k}_rows"], FX[f"adv_{m}_{b}_{k}_scores"],
                                    f"oracle adversarial metric={m} q={b} k={k}")


@pytest.mark.parametrize("k", [60, 1])
def test_rrf_reproduces_vectors(trag, oracle, k):
    dense, sparse = FX["rrf_dense"], FX["rrf_sparse"]
    ids, scores = oracle.rrf(dense, sparse, float(k))
    assert ids.tolist() == FX[f"rrf_{k}_ids"].tolist()
    assert scores.tobytes() == FX[f"rrf_{k}_scores"].tobytes()
    # the shipped host code behind trag_fuse (no GPU involved)
    out = trag.FusionStrategy.RRF(float(k)).fuse([(int(i), 0.0) for i in dense], [(int(i), 0.0) for i in sparse])
    assert [i for i, _ in out] == FX[f"rrf_{k}_ids"].tolist()
    assert np.array([s for _, s in out], dtype=np.float32).tobytes() == FX[f"rrf_{k}_scores"].tobytes()


# ---------------------------------------------------------------- GPU
EXACT, STREAM, MFMA = 1, 2, 3


def make_store(trag, dim, metric, mode):
    return trag.VectorStore(trag.VectorStoreConfig(dimension=dim, metric=trag.DistanceMetric(metric), scan_mode=mode))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, EXACT, STREAM, MFMA])
@pytest.mark.parametrize("d", DIMS)
def test_hip_path_reproduces_config_vectors(trag, d, mode):
    seed = int(FX[f"cfg_{d}_seed"])
    for m in range(3):
        store = make_store(trag, d, m, mode)
        store.fill_synthetic(4096, seed)
        assert store.read_rows(0, 4).tobytes() == FX[f"cfg_{d}_corpus_head"].tobytes()
        qs = make_store(trag, d, m, 0)
        qs.fill_synthetic(8, seed + 100)
        Q = qs.read_rows(0, 8)
        for k in (1, 10, 50, 100):
            for b in range(8):
                rows, scores = store.search_rows(Q[b], k)
                assert_same_ranking(rows, scores, FX[f"cfg_{d}_{m}_rows"][b][:k], FX[f"cfg_{d}_{m}_scores"][b][:k],
                                    f"hip d={d} metric={m} mode={mode} q={b} k={k}")
            rb, sb, nb = store.search_batch_rows(Q, k)      # multi-query pass / matrix cores
            assert nb.tolist() == [k] * 8
            for b in range(8):
                assert_same_ranking(rb[b], sb[b], FX[f"cfg_{d}_{m}_rows"][b][:k], FX[f"cfg_{d}_{m}_scores"][b][:k],
                                    f"hip batch d={d} metric={m} mode={mode} q={b} k={k}")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, EXACT])
def test_hip_path_reproduces_adversarial_vectors(trag, mode):
    corpus, dead, Q = FX["adv_corpus"], FX["adv_dead"], FX["adv_queries"]
    for m in range(3):
        store = make_store(trag, 8, m, mode)
        store.insert_rows(corpus)
        for r in np.nonzero(dead)[0]:
            store.remove_row(int(r))
        for b in range(3):
            for k in (0, 5, 40, 64):
                rows, scores = store.search_rows(Q[b], k)
                assert_same_ranking(rows, scores, FX[f"adv_{m}_{b}_{k}_rows"], FX[f"adv_{m}_{b}_{k}_scores"],
                                    f"hip adversarial metric={m} q={b} k={k} mode={mode}")
