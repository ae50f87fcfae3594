"""index.json (crates/trueno-rag-cli/src/main.rs:133-155): CPU round trip of the
file format; GPU serving of the CLI's query loop against the oracle."""
import json
import os

import numpy as np
import pytest


def make_index(trag, n=50, d=8, seed=0):
    rng = np.random.default_rng(seed)
    emb = rng.standard_normal((n, d)).astype(np.float32)
    emb[3] = 0.0                      # a zero vector scores 0.0
    emb[7] = emb[5]                   # an exact duplicate: tie by index
    emb[9, 0] = np.float32(1e-7)
    emb[9, 1] = np.float32(3.4e20)
    chunks = [trag.PersistedChunk(f"content {i} é\"q\"", f"title {i}" if i % 2 else None, None) for i in range(n)]
    return trag.PersistedIndex(chunks, emb, d, "tfidf", None)


def test_save_load_round_trip(trag, tmp_path):
    p = make_index(trag)
    f = p.save(str(tmp_path / "idx"))
    raw = json.load(open(f, encoding="utf-8"))            # plain JSON, serde field names
    assert set(raw) == {"chunks", "embeddings", "dimension", "embedder_type", "model_name"}
    assert raw["dimension"] == 8 and raw["embedder_type"] == "tfidf" and raw["model_name"] is None
    assert set(raw["chunks"][0]) == {"content", "title", "source"}
    q = trag.PersistedIndex.load(str(tmp_path / "idx"))
    assert q.embeddings.tobytes() == p.embeddings.tobytes()   # every f32 survives its shortest decimal form
    assert [c.content for c in q.chunks] == [c.content for c in p.chunks]
    assert [c.title for c in q.chunks] == [c.title for c in p.chunks]


def test_load_defaults_and_errors(trag, tmp_path):
    d = tmp_path / "i2"
    os.makedirs(d)
    json.dump({"chunks": [{"content": "a", "title": None, "source": None}], "embeddings": [[1.0, 2.0]],
               "dimension": 2}, open(d / "index.json", "w"))
    p = trag.PersistedIndex.load(str(d))
    assert p.embedder_type == "" and p.model_name is None    # #[serde(default)]
    with pytest.raises(FileNotFoundError):
        trag.PersistedIndex.load(str(tmp_path / "missing"))
    json.dump({"chunks": [{"content": "a", "title": None, "source": None}], "embeddings": [[1.0]],
               "dimension": 2}, open(d / "index.json", "w"))
    with pytest.raises(trag.DimensionMismatch):
        trag.PersistedIndex.load(str(d))


@pytest.mark.gpu
def test_query_loop_on_gpu(trag, oracle, tmp_path):
    p = make_index(trag, n=3000, d=64, seed=2)
    p.save(str(tmp_path / "idx"))
    srv = trag.IndexServer.open(str(tmp_path / "idx"))
    rng = np.random.default_rng(5)
    for _ in range(3):
        q = rng.standard_normal(64).astype(np.float32)
        got = srv.query(q, 10)
        er, es = oracle.search(p.embeddings, q, 10, 0)
        assert [g[0] for g in got] == er.tolist()
        assert np.asarray([g[1] for g in got], dtype=np.float32).tobytes() == es.tobytes()
    got = srv.query(p.embeddings[5], 3)
    assert [g[0] for g in got][:2] == [5, 7]              # duplicate rows: lower index first
    recs = srv.results_json(p.embeddings[5], 2)
    assert recs[0]["rank"] == 1 and recs[0]["content"] == p.chunks[5].content and recs[0]["title"] == "title 5"
    assert srv.query(np.ones(3, dtype=np.float32), 4) == [(0, 0.0), (1, 0.0), (2, 0.0), (3, 0.0)]
