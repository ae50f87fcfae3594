/*
 * trag_oracle_bm25.c -- TEST INFRASTRUCTURE ONLY. Not part of the product.
 *
 * A plain-C restatement of trueno-rag's BM25 sparse index over documents that
 * are already tokenised into term ids (the tokenizer itself, src/index.rs:111-124,
 * is string work and is restated in Python next to the tests that pin it).
 *
 * Parity status: pinned only by the reference's inline unit tests
 * (src/index.rs:547-665: result counts, "more occurrences ranks first", empty /
 * stop-word-only / unmatched queries, removal), transcribed in
 * tests/golden/ref_bm25_vectors.json. The reference holds no numeric BM25
 * golden values: exact scores are "parity unpinned" beyond the formula below.
 *
 * Reference lines followed (paths relative to /root/reference):
 *   src/index.rs:176-204  BM25Index::add        -> doc length = token count, tf per term,
 *                                                   df per term, avg = total / count
 *   src/index.rs:136-154  score_term            -> bm25_term_score
 *   src/index.rs:212-243  BM25Index::search     -> oracle_bm25_search
 *   src/index.rs:245-275  BM25Index::remove     -> the `removed` mask: a removed document
 *                                                   leaves no trace in df, count or avg
 *
 * Arithmetic: all f32, one rounding per operation, in the reference's order:
 *   idf     = ln((n - df + 0.5) / (df + 0.5) + 1.0)
 *   tf_norm = (tf * (k1 + 1.0)) / (tf + k1 * ((1.0 - b) + (b * len) / avg))
 *   score   = sum over the query's terms, in query order, duplicates included,
 *             of idf * tf_norm (0.0 for a term the document lacks)
 * ln is the C library's logf, which is what Rust's f32::ln calls on Linux.
 *
 * Tie rule: the reference collects candidates in a HashSet (random order) and
 * stable-sorts; here equal scores are ordered by ascending document index.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static float bm25_term_score(float tf, float df, float n, float doc_len, float avg, float k1, float b) {
    if (tf == 0.0f) return 0.0f;
    const float idf = logf((n - df + 0.5f) / (df + 0.5f) + 1.0f);
    const float tf_norm = (tf * (k1 + 1.0f)) / (tf + k1 * (1.0f - b + b * doc_len / avg));
    return idf * tf_norm;
}

typedef struct { uint32_t doc; float score; } hit_t;

static void merge_sort_hits(hit_t *a, hit_t *tmp, size_t n) {
    if (n < 2) return;
    const size_t h = n / 2;
    merge_sort_hits(a, tmp, h);
    merge_sort_hits(a + h, tmp, n - h);
    size_t i = 0, j = h, o = 0;
    while (i < h && j < n) {
        /* stable, descending: take the right element only when it is strictly greater */
        if (a[j].score > a[i].score) tmp[o++] = a[j++];
        else tmp[o++] = a[i++];
    }
    while (i < h) tmp[o++] = a[i++];
    while (j < n) tmp[o++] = a[j++];
    memcpy(a, tmp, n * sizeof(hit_t));
}

/* terms: concatenated token ids; document i owns terms[doc_off[i] .. doc_off[i+1]).
 * removed: one byte per document (nonzero = removed), or NULL.
 * Returns the number of results written (<= k). */
uint32_t oracle_bm25_search(const uint32_t *terms, const uint64_t *doc_off, uint64_t n_docs,
                            const uint8_t *removed, float k1, float b, const uint32_t *q_terms,
                            uint32_t nq, uint32_t k, uint32_t *out_docs, float *out_scores) {
    if (nq == 0 || k == 0) return 0;
    uint64_t count = 0, total = 0;
    for (uint64_t i = 0; i < n_docs; ++i) {
        if (removed && removed[i]) continue;
        count += 1;
        total += doc_off[i + 1] - doc_off[i];
    }
    if (count == 0) return 0;
    const float n = (float)(uint32_t)count;
    const float avg = (float)(uint32_t)total / (float)(uint32_t)count;
    /* document frequency of every query term */
    float *df = (float *)calloc(nq, sizeof(float));
    uint32_t *dfc = (uint32_t *)calloc(nq, sizeof(uint32_t));
    uint32_t *tf = (uint32_t *)calloc(nq, sizeof(uint32_t));
    hit_t *hits = (hit_t *)malloc((size_t)(n_docs ? n_docs : 1) * sizeof(hit_t));
    size_t n_hits = 0;
    for (uint64_t i = 0; i < n_docs; ++i) {
        if (removed && removed[i]) continue;
        for (uint32_t t = 0; t < nq; ++t) {
            for (uint64_t p = doc_off[i]; p < doc_off[i + 1]; ++p)
                if (terms[p] == q_terms[t]) { dfc[t] += 1; break; }
        }
    }
    for (uint32_t t = 0; t < nq; ++t) df[t] = (float)dfc[t];
    for (uint64_t i = 0; i < n_docs; ++i) {
        if (removed && removed[i]) continue;
        int any = 0;
        for (uint32_t t = 0; t < nq; ++t) {
            uint32_t c = 0;
            for (uint64_t p = doc_off[i]; p < doc_off[i + 1]; ++p) c += terms[p] == q_terms[t];
            tf[t] = c;
            any |= c != 0;
        }
        if (!any) continue; /* not a candidate: shares no term with the query */
        const float len = (float)(uint32_t)(doc_off[i + 1] - doc_off[i]);
        float score = 0.0f;
        for (uint32_t t = 0; t < nq; ++t)
            score = score + bm25_term_score((float)tf[t], df[t], n, len, avg, k1, b);
        if (score > 0.0f) {
            hits[n_hits].doc = (uint32_t)i;
            hits[n_hits].score = score;
            n_hits += 1;
        }
    }
    hit_t *tmp = (hit_t *)malloc((n_hits ? n_hits : 1) * sizeof(hit_t));
    merge_sort_hits(hits, tmp, n_hits);
    const uint32_t n_out = n_hits < k ? (uint32_t)n_hits : k;
    for (uint32_t i = 0; i < n_out; ++i) {
        out_docs[i] = hits[i].doc;
        out_scores[i] = hits[i].score;
    }
    free(tmp);
    free(hits);
    free(tf);
    free(dfc);
    free(df);
    return n_out;
}

/* Synthetic documents: token `pos` of document `doc` is drawn from Zipf(1.0)
 * over `vocab` terms (term 0 the most frequent) by inverting a float64 CDF
 * with a 53-bit uniform from a counter hash of (seed, doc, pos). */
static inline uint64_t bm25_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

void oracle_bm25_synth_docs(uint64_t seed, uint64_t doc0, uint64_t n_docs, uint32_t vocab,
                            uint32_t tokens_per_doc, uint32_t *out_terms) {
    double *cdf = (double *)malloc((size_t)vocab * sizeof(double));
    double acc = 0.0;
    for (uint32_t r = 0; r < vocab; ++r) {
        acc += 1.0 / (double)(r + 1);
        cdf[r] = acc;
    }
    const double total = acc;
    for (uint64_t i = 0; i < n_docs; ++i) {
        for (uint32_t p = 0; p < tokens_per_doc; ++p) {
            const uint64_t h = bm25_splitmix64(seed ^ bm25_splitmix64((doc0 + i) * 0x100000001B3ull + p));
            const double u = (double)(h >> 11) * (1.0 / 9007199254740992.0) * total;
            uint32_t lo = 0, hi = vocab - 1; /* first r with cdf[r] > u */
            while (lo < hi) {
                const uint32_t mid = lo + (hi - lo) / 2;
                if (cdf[mid] > u) hi = mid;
                else lo = mid + 1;
            }
            out_terms[i * tokens_per_doc + p] = lo;
        }
    }
    free(cdf);
}
