"""CPU: the BM25 oracle against the reference's own unit tests (transcribed in
tests/golden/ref_bm25_vectors.json) and against an independent float64
evaluation of the published formula."""
import json
import os
import random
import string

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "ref_bm25_vectors.json")))


def docs_of(case):
    if "docs_rule" in case:
        return [f"document {i} about rust" for i in range(10)]
    return case["docs"]


@pytest.fixture(scope="module")
def text():
    from oracle import bm25_text
    return bm25_text


@pytest.mark.parametrize("case", GOLD["tokenize"], ids=lambda c: c["name"])
def test_tokenize(text, case):
    toks = text.tokenize(case["text"])
    for t in case["contains"]:
        assert t in toks
    for t in case["not_contains"]:
        assert t not in toks


def test_tokenize_rules(text):
    # split at every non-alphanumeric char, 1-byte tokens dropped, stop words after lowercasing
    assert text.tokenize("x y2 THE Rust-lang_c0de a1") == ["y2", "rust", "lang", "c0de", "a1"]
    assert text.tokenize("é ab") == ["é", "ab"]  # 2 bytes in UTF-8 passes the length filter


@pytest.mark.parametrize("case", GOLD["add"], ids=lambda c: c["name"])
def test_add(text, case):
    ix = text.TextIndex()
    for d in case["docs"]:
        ix.add(d)
    assert len(ix) == case["expect_len"]
    for t in case["indexed_terms"]:
        assert t in ix.vocab


@pytest.mark.parametrize("case", GOLD["search"], ids=lambda c: c["name"])
def test_search(text, case):
    ix = text.TextIndex()
    for d in docs_of(case):
        ix.add(d)
    docs, scores = ix.search(case["query"], case["k"])
    if "expect_len" in case:
        assert len(docs) == case["expect_len"]
    if case.get("expect_nonempty"):
        assert len(docs) > 0
    for d in case.get("expect_contains", []):
        assert d in docs.tolist()
    if "expect_first" in case:
        assert docs[0] == case["expect_first"]
    assert all(scores[i] >= scores[i + 1] for i in range(len(scores) - 1))
    assert all(s > 0 for s in scores)


@pytest.mark.parametrize("case", GOLD["remove"], ids=lambda c: c["name"])
def test_remove(text, case):
    ix = text.TextIndex()
    for d in case["docs"]:
        ix.add(d)
    for r in case["remove"]:
        ix.remove(r)
    assert len(ix) == case["expect_len_after"]
    assert len(ix.search(case["query"], case["k"])[0]) == case["expect_len"]


def test_avg_doc_length(text):
    case = GOLD["avg_doc_length"][0]
    ix = text.TextIndex()
    for d in case["docs"]:
        ix.add(d)
    assert ix.avg_doc_length() > 0.0
    # "short text" -> 2 tokens; the longer one -> longer, piece, text, about, programming
    assert ix.avg_doc_length() == 3.5


def test_properties(text):
    rnd = random.Random(5)
    for _ in range(50):
        ix = text.TextIndex()
        n = rnd.randint(3, 19)
        for _ in range(n):
            ix.add("".join(rnd.choice(string.ascii_letters + " ") for _ in range(rnd.randint(5, 50))))
        assert len(ix) == n
        k = rnd.randint(1, 9)
        assert len(ix.search("test", k)[0]) <= k
        q = "".join(rnd.choice(string.ascii_letters) for _ in range(rnd.randint(3, 10)))
        assert all(s >= 0 for s in ix.search(q, 100)[1])


def test_formula_float64(oracle):
    """Scores agree with a float64 evaluation of ln((N-df+.5)/(df+.5)+1) * tf(k1+1)/(tf+k1(1-b+b*len/avg))."""
    rng = np.random.default_rng(3)
    n_docs, vocab = 300, 40
    lens = rng.integers(0, 30, n_docs)
    off = np.zeros(n_docs + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    terms = rng.integers(0, vocab, int(off[-1])).astype(np.uint32)
    q = np.array([3, 7, 7, 39, 100], dtype=np.uint32)
    k1, b = 1.2, 0.75
    docs, scores = oracle.bm25_search(terms, off, q, n_docs, k1, b)
    avg = lens.sum() / n_docs
    want = {}
    for d in range(n_docs):
        toks = terms[int(off[d]):int(off[d + 1])]
        s = 0.0
        for t in q:
            tf = int((toks == t).sum())
            if tf == 0:
                continue
            df = sum(1 for e in range(n_docs) if (terms[int(off[e]):int(off[e + 1])] == t).any())
            idf = np.log((n_docs - df + 0.5) / (df + 0.5) + 1.0)
            s += idf * tf * (k1 + 1) / (tf + k1 * (1 - b + b * len(toks) / avg))
        if s > 0:
            want[d] = s
    assert sorted(docs.tolist()) == sorted(want)
    for d, s in zip(docs, scores):
        assert abs(s - want[int(d)]) <= 1e-5 * want[int(d)]
    assert all(scores[i] >= scores[i + 1] for i in range(len(scores) - 1))


def test_removed_docs_leave_no_trace(oracle):
    toks = oracle.bm25_synth_docs(11, 0, 400, 60, 12)
    off = np.arange(401, dtype=np.uint64) * 12
    removed = np.zeros(400, dtype=np.uint8)
    removed[::3] = 1
    q = [0, 5, 17]
    d1, s1 = oracle.bm25_search(toks.ravel(), off, q, 50, removed=removed)
    keep = np.flatnonzero(removed == 0)
    d2, s2 = oracle.bm25_search(toks[keep].ravel(), np.arange(len(keep) + 1, dtype=np.uint64) * 12, q, 50)
    assert keep[d2].tolist() == d1.tolist()
    assert s1.tobytes() == s2.tobytes()


def test_synth_docs_zipf(oracle):
    toks = oracle.bm25_synth_docs(1005, 0, 2000, 1000, 64)
    assert toks.shape == (2000, 64) and toks.max() < 1000
    c = np.bincount(toks.ravel(), minlength=1000).astype(float)
    # Zipf(1): count(r) ~ 1/(r+1)
    assert 1.6 < c[0] / c[1] < 2.4 and 2.4 < c[0] / c[2] < 3.6
    again = oracle.bm25_synth_docs(1005, 100, 10, 1000, 64)
    assert (again == toks[100:110]).all()
