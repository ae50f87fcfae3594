// CPU-only driver for the host code that needs no GPU -- fusion and list merge (trag_fuse.cpp),
// MockEmbedder (trag_embed.cpp), the error string (trag_error.cpp) -- linked with the C oracle and
// built with -fsanitize=address,undefined (make -C trueno-rag_amd/csrc asan). It checks the host
// code against the oracle on seeded inputs, including the degenerate ones (empty lists, duplicate
// ids, NaN scores, ragged merges), so that the sanitizers see every branch. Exit code 0 = clean.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "trag.h"

extern "C" {
uint32_t oracle_fuse(int strategy, float param, const uint64_t* dense_ids, const float* dense_scores, uint32_t nd,
                     const uint64_t* sparse_ids, const float* sparse_scores, uint32_t ns, uint64_t* out_ids,
                     float* out_scores);
uint32_t oracle_search(const float* corpus, const uint8_t* dead, uint64_t n, uint32_t d, int metric, const float* q,
                       uint32_t k, uint32_t* out_rows, float* out_scores);
void oracle_synth_rows(uint64_t seed, uint64_t row0, uint64_t n, uint32_t d, int normalize, float* out);
uint32_t oracle_bm25_search(const uint32_t* terms, const uint64_t* doc_off, uint64_t n_docs, const uint8_t* removed,
                            float k1, float b, const uint32_t* q_terms, uint32_t nq, uint32_t k, uint32_t* out_docs,
                            float* out_scores);
void oracle_bm25_synth_docs(uint64_t seed, uint64_t doc0, uint64_t n_docs, uint32_t vocab, uint32_t tokens_per_doc,
                            uint32_t* out_terms);
}

static int failures = 0;
#define EXPECT(cond)                                                                    \
    do {                                                                                \
        if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } \
    } while (0)

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}

static bool same_bits(float a, float b) { return std::memcmp(&a, &b, 4) == 0 || (a == 0.0f && b == 0.0f); }

int main() {
    // ---- fusion: all six strategies against the oracle, random overlapping lists with duplicates
    for (int round = 0; round < 400; ++round) {
        const uint32_t nd = (uint32_t)(rnd() % 60), ns = (uint32_t)(rnd() % 60);
        std::vector<uint64_t> di(nd), si(ns);
        std::vector<float> dsc(nd), ssc(ns);
        for (uint32_t i = 0; i < nd; ++i) { di[i] = rnd() % 70; dsc[i] = (float)(rnd() % 1000) / 37.0f - 5.0f; }
        for (uint32_t i = 0; i < ns; ++i) { si[i] = rnd() % 70; ssc[i] = (float)(rnd() % 1000) / 91.0f; }
        if (round % 50 == 7 && nd) dsc[0] = NAN;
        if (round % 50 == 9) { for (auto& v : dsc) v = 2.5f; }   // zero range / zero deviation
        const int strategy = round % 6;
        const float param = strategy == 0 ? (round % 2 ? 60.0f : 1.0f) : 0.3f;
        std::vector<uint64_t> oi(nd + ns + 1), gi(nd + ns + 1);
        std::vector<float> os(nd + ns + 1), gs(nd + ns + 1);
        const uint32_t on = oracle_fuse(strategy, param, di.data(), dsc.data(), nd, si.data(), ssc.data(), ns, oi.data(), os.data());
        uint32_t gn = 12345;
        const trag_status st = trag_fuse((uint32_t)strategy, param, di.data(), dsc.data(), nd, si.data(), ssc.data(), ns,
                                         gi.data(), gs.data(), &gn);
        EXPECT(st == TRAG_OK);
        EXPECT(gn == on);
        for (uint32_t i = 0; i < on && i < gn; ++i) {
            EXPECT(gi[i] == oi[i]);
            EXPECT(same_bits(gs[i], os[i]) || (gs[i] != gs[i] && os[i] != os[i]));
        }
    }
    {   // the reference's own known answer: 1/61 + 1/61 (src/fusion.rs:314-325)
        const uint64_t a[1] = {1}, b[1] = {1};
        uint64_t oi[2];
        float os[2];
        uint32_t n = 0;
        EXPECT(trag_fuse_rrf(a, 1, b, 1, 60.0f, oi, os, &n) == TRAG_OK && n == 1 && std::fabs(os[0] - 2.0f / 61.0f) < 1e-6f);
        EXPECT(trag_fuse_rrf(nullptr, 0, nullptr, 0, 60.0f, nullptr, nullptr, &n) == TRAG_OK && n == 0);
    }
    {   // failures carry their own message
        uint32_t n = 5;
        const uint64_t a[1] = {1};
        uint64_t oi[2];
        float os[2];
        EXPECT(trag_fuse(9, 1.0f, a, nullptr, 1, a, nullptr, 1, oi, os, &n) == TRAG_INVALID_ARG);
        EXPECT(std::strstr(trag_last_error(), "unknown strategy") != nullptr);
        EXPECT(trag_fuse(3, 1.0f, a, nullptr, 1, a, nullptr, 1, oi, os, &n) == TRAG_INVALID_ARG);
        EXPECT(std::strstr(trag_last_error(), "scores") != nullptr);
        EXPECT(trag_merge_topk(nullptr, nullptr, nullptr, 3, 4, 4, nullptr, nullptr, &n) == TRAG_INVALID_ARG);
        EXPECT(std::strstr(trag_last_error(), "trag_merge_topk") != nullptr);
        EXPECT(trag_mock_embed(nullptr, 3, 8, 1, os) == TRAG_INVALID_ARG);
    }
    // ---- merge of per-shard lists against one global oracle search
    for (int round = 0; round < 40; ++round) {
        const uint32_t n = 300 + (uint32_t)(rnd() % 700), d = 8, shards = 1 + (uint32_t)(rnd() % 5), k = 1 + (uint32_t)(rnd() % 20);
        std::vector<float> corpus((size_t)n * d), q(d);
        oracle_synth_rows(100 + round, 0, n, d, 1, corpus.data());
        oracle_synth_rows(900 + round, 0, 1, d, 1, q.data());
        for (uint32_t c = 0; c < d; ++c) corpus[(size_t)7 * d + c] = corpus[(size_t)(n - 1) * d + c];   // a tie across shards
        std::vector<uint32_t> er(k), l_rows((size_t)shards * k), l_n(shards), gr(k);
        std::vector<float> es(k), l_scores((size_t)shards * k), gs(k);
        const uint32_t en = oracle_search(corpus.data(), nullptr, n, d, round % 3, q.data(), k, er.data(), es.data());
        const uint32_t per = (n + shards - 1) / shards;
        for (uint32_t s = 0; s < shards; ++s) {
            const uint32_t r0 = s * per, cnt = r0 < n ? (n - r0 < per ? n - r0 : per) : 0;
            l_n[s] = cnt ? oracle_search(corpus.data() + (size_t)r0 * d, nullptr, cnt, d, round % 3, q.data(), k,
                                         l_rows.data() + (size_t)s * k, l_scores.data() + (size_t)s * k) : 0;
            for (uint32_t i = 0; i < l_n[s]; ++i) l_rows[(size_t)s * k + i] += r0;
        }
        uint32_t gn = 0;
        EXPECT(trag_merge_topk(l_rows.data(), l_scores.data(), l_n.data(), shards, k, k, gr.data(), gs.data(), &gn) == TRAG_OK);
        EXPECT(gn == en);
        for (uint32_t i = 0; i < en && i < gn; ++i) EXPECT(gr[i] == er[i] && same_bits(gs[i], es[i]));
    }
    // ---- MockEmbedder: the facts the reference pins (src/embed.rs:992-1036)
    {
        std::vector<float> a(384), b(384), c(384), raw(16);
        EXPECT(trag_mock_embed("hello world", 11, 384, 1, a.data()) == TRAG_OK);
        EXPECT(trag_mock_embed("hello world", 11, 384, 1, b.data()) == TRAG_OK);
        EXPECT(trag_mock_embed("hello worle", 11, 384, 1, c.data()) == TRAG_OK);
        float ss = 0.0f;
        bool same = true, differs = false;
        for (int i = 0; i < 384; ++i) { ss += a[i] * a[i]; same = same && a[i] == b[i]; differs = differs || a[i] != c[i]; }
        EXPECT(std::fabs(std::sqrt(ss) - 1.0f) < 1e-5f && same && differs);
        EXPECT(trag_mock_embed("", 0, 16, 0, raw.data()) == TRAG_OK);
        for (float v : raw) EXPECT(v >= -1.0f && v <= 1.0f);
        EXPECT(trag_mock_embed("x", 1, 0, 1, nullptr) == TRAG_OK);
    }
    // ---- the BM25 oracle on synthetic documents (out-of-bounds reads would show here)
    {
        const uint32_t n_docs = 500, vocab = 200, tpd = 17;
        std::vector<uint32_t> terms((size_t)n_docs * tpd), docs(64);
        std::vector<uint64_t> off(n_docs + 1);
        std::vector<float> sc(64);
        oracle_bm25_synth_docs(5, 0, n_docs, vocab, tpd, terms.data());
        for (uint32_t i = 0; i <= n_docs; ++i) off[i] = (uint64_t)i * tpd;
        std::vector<uint8_t> removed(n_docs, 0);
        removed[3] = removed[499] = 1;
        const uint32_t qt[5] = {0, 7, 7, 199, 4000};
        const uint32_t n = oracle_bm25_search(terms.data(), off.data(), n_docs, removed.data(), 1.2f, 0.75f, qt, 5, 64, docs.data(), sc.data());
        EXPECT(n > 0 && n <= 64);
        for (uint32_t i = 1; i < n; ++i) EXPECT(sc[i - 1] >= sc[i] && docs[i] != 3 && docs[i] != 499);
    }
    std::printf(failures ? "%d FAILURES\n" : "sanitized host code: all checks passed\n", failures);
    return failures ? 1 : 0;
}
