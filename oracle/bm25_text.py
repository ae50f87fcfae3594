"""Text front end of the BM25 oracle -- TEST INFRASTRUCTURE ONLY.

Restates BM25Index::tokenize (src/index.rs:111-124 of the reference) and its
default stop-word list (src/index.rs:93-108) in Python, assigns term ids in
first-seen order and runs the C oracle (oracle_bm25_search) over the ids.
Pinned by the reference's tokenizer / search unit tests transcribed in
tests/golden/ref_bm25_vectors.json; str.isalnum() stands in for Rust's
char::is_alphanumeric (identical on ASCII, which is all the reference tests use).
"""
import numpy as np

from . import oracle

STOPWORDS = set("""
a an the is are was were be been being have has had do does did will would could should may might must
shall can need dare ought used to of in for on with at by from as into through during before after above
below between under again further then once here there when where why how all each few more most other some
such no nor not only own same so than too very just and but if or because until while this that these
those it its
""".split())


def tokenize(text, stopwords=STOPWORDS, lowercase=True):
    pieces, cur = [], []
    for ch in text:
        if ch.isalnum():
            cur.append(ch)
        else:
            if cur:
                pieces.append("".join(cur))
            cur = []
    if cur:
        pieces.append("".join(cur))
    if lowercase:
        pieces = [p.lower() for p in pieces]
    return [p for p in pieces if p not in stopwords and len(p.encode("utf-8")) >= 2]


class TextIndex:
    """Documents as text; search through the C oracle."""

    def __init__(self, k1=1.2, b=0.75):
        self.k1, self.b = k1, b
        self.vocab = {}
        self.docs = []       # list of np.uint32 arrays
        self.removed = []

    def add(self, text):
        ids = []
        for t in tokenize(text):
            ids.append(self.vocab.setdefault(t, len(self.vocab)))
        self.docs.append(np.asarray(ids, dtype=np.uint32))
        self.removed.append(0)
        return len(self.docs) - 1

    def remove(self, doc):
        self.removed[doc] = 1

    def __len__(self):
        return len(self.docs) - sum(self.removed)

    def avg_doc_length(self):
        live = [len(d) for d, r in zip(self.docs, self.removed) if not r]
        return float(np.float32(sum(live)) / np.float32(len(live))) if live else 0.0

    def csr(self):
        off = np.zeros(len(self.docs) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(d) for d in self.docs])
        terms = np.concatenate(self.docs) if len(self.docs) and off[-1] else np.zeros(0, dtype=np.uint32)
        return terms, off

    def search(self, query, k):
        q = tokenize(query)
        if not q:
            return np.zeros(0, np.uint32), np.zeros(0, np.float32)
        unknown = 0x0FFFFFFF
        qi = np.asarray([self.vocab.get(t, unknown) for t in q], dtype=np.uint32)
        terms, off = self.csr()
        return oracle.bm25_search(terms, off, qi, k, self.k1, self.b, np.asarray(self.removed, dtype=np.uint8))
