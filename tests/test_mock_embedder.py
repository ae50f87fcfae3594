"""CPU: MockEmbedder (host code of the library, no GPU needed) against the
reference's unit tests (src/embed.rs:985-1090) and against the independent
pure-Python restatement in oracle/mock_embed.py. Exact values are parity
unpinned: the reference pins length, unit norm, determinism and sensitivity."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def me():
    from oracle import mock_embed
    return mock_embed


def test_embed_len_and_norm(trag):
    e = trag.MockEmbedder.new(128)
    v = e.embed("Hello world")
    assert len(v) == 128
    s = np.float32(0)
    for x in v:
        s = np.float32(s + np.float32(x * x))
    assert abs(float(np.sqrt(s)) - 1.0) < 1e-5          # src/embed.rs:992-1000


def test_embed_empty_is_error(trag):
    e = trag.MockEmbedder.new(128)
    with pytest.raises(trag.EmptyDocument):
        e.embed("")                                       # :1003-1007
    with pytest.raises(trag.QueryError):
        e.embed_query("")                                 # :184-187


def test_embed_batch_and_model_id(trag):
    e = trag.MockEmbedder.new(64)
    embs = e.embed_batch(["Hello", "World", "Test"])
    assert len(embs) == 3 and all(len(x) == 64 for x in embs)   # :1010-1019
    assert e.dimension() == 64 and e.model_id() == "mock-embedder"
    assert trag.MockEmbedder.new(768).with_model_id("custom-model").model_id() == "custom-model"


def test_deterministic_and_text_sensitive(trag):
    e = trag.MockEmbedder.new(128)
    assert e.embed("Hello").tobytes() == e.embed("Hello").tobytes()      # :1022-1027
    assert e.embed("Hello").tobytes() != e.embed("World").tobytes()      # :1030-1035


def test_query_prefix(trag):
    e = trag.MockEmbedder.new(128).with_config(trag.EmbeddingConfig(query_prefix="query: "))
    assert e.embed_query("test").tobytes() != e.embed_document("test").tobytes()   # :1038-1050
    assert e.embed_query("test").tobytes() == trag.MockEmbedder.new(128).embed("query: test").tobytes()


def test_embed_chunks(trag):
    e = trag.MockEmbedder.new(64)
    chunks = [trag.Chunk(content="First chunk"), trag.Chunk(content="Second chunk")]
    e.embed_chunks(chunks)
    assert all(c.embedding is not None and len(c.embedding) == 64 for c in chunks)   # :1053-1070


def test_no_normalize(trag):
    e = trag.MockEmbedder.new(128).with_config(trag.EmbeddingConfig(normalize=False))
    v = e.embed("test")
    assert np.all(v >= -1.0) and np.all(v <= 1.0)
    assert abs(float(np.linalg.norm(v)) - 1.0) > 1e-3


@pytest.mark.parametrize("text,dim", [("Hello world", 128), ("a", 7), ("document 17", 384), ("naïve café", 33),
                                       ("x" * 100, 16)])
def test_matches_python_restatement(trag, me, text, dim):
    for norm in (True, False):
        got = trag.MockEmbedder.new(dim).with_config(trag.EmbeddingConfig(normalize=norm)).embed(text)
        assert got.tobytes() == me.hash_to_vector(text, dim, norm).tobytes()


def test_siphash_properties(me):
    # not a published test vector (none is available offline): structural checks only
    assert me.siphash13(b"") != me.siphash13(b"\x00")
    assert me.siphash13(b"abcdefgh") != me.siphash13(b"abcdefgi")
    assert 0 <= me.siphash13(b"abc") < 1 << 64
