"""CPU: LexicalReranker / NoOpReranker mirror against the reference's unit tests
(src/rerank.rs:305-392, 459-490)."""


def results(trag, texts):
    return [trag.RetrievalResult(trag.Chunk(content=t)) for t in texts]


def test_defaults_and_weights(trag):
    r = trag.LexicalReranker.new()
    assert abs(r.exact_match_weight - 0.3) < 0.01 and abs(r.coverage_weight - 0.5) < 0.01
    assert abs(r.position_weight - 0.2) < 0.01 and r.case_insensitive
    w = trag.LexicalReranker.new().with_weights(0.5, 0.3, 0.2)
    assert abs(w.exact_match_weight - 0.5) < 0.01


def test_exact_match_scores_higher(trag):
    r = trag.LexicalReranker.new()
    out = r.rerank("machine learning", results(trag, ["This contains the exact query machine learning",
                                                      "This mentions machine and learning separately"]), 2)
    assert out[0].rerank_score > out[1].rerank_score
    assert out[0].chunk.content.startswith("This contains")


def test_coverage(trag):
    r = trag.LexicalReranker.new()
    out = r.rerank("machine learning neural networks", results(trag, ["machine learning algorithms",
                                                                      "machine only here"]), 2)
    assert out[0].rerank_score >= out[1].rerank_score


def test_top_k_and_empty_query(trag):
    r = trag.LexicalReranker.new()
    assert len(r.rerank("doc", results(trag, [f"doc {i}" for i in range(10)]), 3)) == 3
    out = r.rerank("", results(trag, ["test content"]), 10)
    assert len(out) == 1 and abs(out[0].rerank_score) < 0.001


def test_case_insensitive(trag):
    r = trag.LexicalReranker.new()
    out = r.rerank("Machine Learning", results(trag, ["MACHINE LEARNING", "machine learning"]), 2)
    assert abs(out[0].rerank_score - out[1].rerank_score) < 0.01


def test_score_value(trag):
    # "machine learning" in "x machine learning": exact 1, coverage 1, positions 2 and 10
    r = trag.LexicalReranker.new()
    want = 0.3 + 0.5 + 0.2 * ((1 / 1.02 + 1 / 1.10) / 2)
    assert abs(r.score("machine learning", "x machine learning") - want) < 1e-6


def test_noop(trag):
    cands = results(trag, ["first", "second", "third"])
    out = trag.NoOpReranker().rerank("query", cands, 2)
    assert [c.chunk.content for c in out] == ["first", "second"]
    assert all(c.rerank_score is None for c in out)
