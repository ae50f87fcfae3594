// The reference's VectorStore / fusion / retriever unit tests (src/index.rs:667-867,
// src/fusion.rs:267-325, src/retrieve.rs:464-531) driven through the C++ host mirror.
// Needs a GPU. Exit code 0 = all passed.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "trueno_rag.hpp"

using namespace trueno_rag;

static int failures = 0;
#define EXPECT(cond)                                                                    \
    do {                                                                                \
        if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } \
    } while (0)

static Chunk chunk_with(const std::string& text, std::vector<float> e) { return Chunk(text).with_embedding(std::move(e)); }

struct StubEmbedder {
    std::vector<float> embed_query(const std::string& q) const {
        if (q.empty()) throw QueryError("empty query");
        std::vector<float> v(3, 0.0f);
        v[(q[0] - 'a') % 3] = 1.0f;
        return v;
    }
};
struct StubSparse {
    std::vector<ChunkId> ids;
    void add(const Chunk& c) { ids.push_back(c.id); }
    std::vector<std::pair<ChunkId, float>> search(const std::string&, size_t k) const {
        std::vector<std::pair<ChunkId, float>> out;
        for (size_t i = 0; i < ids.size() && i < k; ++i) out.emplace_back(ids[ids.size() - 1 - i], 5.0f - (float)i);
        return out;
    }
};

int main() {
    {   // test_vector_store_new / insert / remove
        VectorStore store = VectorStore::with_dimension(384);
        EXPECT(store.config().dimension == 384 && store.is_empty());
        VectorStore s3 = VectorStore::with_dimension(3);
        Chunk c = chunk_with("test", {1.0f, 0.0f, 0.0f});
        ChunkId id = c.id;
        s3.insert(c);
        EXPECT(s3.len() == 1 && !s3.is_empty() && s3.get(id) != nullptr);
        EXPECT(s3.remove(id).has_value() && s3.len() == 0 && s3.get(id) == nullptr);
        EXPECT(!s3.remove(ChunkId::random()).has_value());
    }
    {   // errors: src/index.rs:704-726, 793-797
        VectorStore s = VectorStore::with_dimension(3);
        try { s.insert(Chunk("no embedding")); EXPECT(false); } catch (const InvalidConfig&) {}
        try { s.insert(chunk_with("x", {1.0f, 0.0f})); EXPECT(false); }
        catch (const DimensionMismatch& e) { EXPECT(e.expected == 3 && e.actual == 2); }
        try { s.search({1.0f, 0.0f}, 10); EXPECT(false); }
        catch (const DimensionMismatch& e) { EXPECT(e.expected == 3 && e.actual == 2); }
    }
    {   // test_vector_store_search_cosine, :742-772
        VectorStore s = VectorStore::with_dimension(3);
        Chunk c1 = chunk_with("north", {1.0f, 0.0f, 0.0f}), c2 = chunk_with("east", {0.0f, 1.0f, 0.0f});
        Chunk c3 = chunk_with("diagonal", {0.70710677f, 0.70710677f, 0.0f});
        ChunkId id1 = c1.id, id3 = c3.id;
        s.insert(c1); s.insert(c2); s.insert(c3);
        auto r = s.search({0.9f, 0.1f, 0.0f}, 10);
        EXPECT(r.size() == 3 && r[0].first == id1 && r[1].first == id3);
    }
    {   // not in the reference: the optional bf16 copy never changes a result (small store: exact path)
        VectorStore s = VectorStore::with_dimension(8);
        s.set_bf16_mirror(true);
        EXPECT(s.has_bf16_mirror());
        Chunk c1 = chunk_with("a", {1, 0, 0, 0, 0, 0, 0, 0}), c2 = chunk_with("b", {0, 1, 0, 0, 0, 0, 0, 0});
        ChunkId id2 = c2.id;
        s.insert(c1); s.insert(c2);
        auto r = s.search({0.1f, 0.9f, 0, 0, 0, 0, 0, 0}, 1);
        EXPECT(r.size() == 1 && r[0].first == id2);
        s.set_bf16_mirror(false);
        EXPECT(!s.has_bf16_mirror());
        VectorStore s3 = VectorStore::with_dimension(3);
        try { s3.set_bf16_mirror(true); EXPECT(false); } catch (const std::exception&) {}
    }
    {   // test_vector_store_search_top_k, :775-790
        VectorStore s = VectorStore::with_dimension(3);
        for (int i = 0; i < 10; ++i) s.insert(chunk_with("chunk", {(float)i, 0.0f, 0.0f}));
        EXPECT(s.search({9.0f, 0.0f, 0.0f}, 3).size() == 3);
    }
    {   // euclidean / dot product, :822-867
        VectorStoreConfig cfg; cfg.dimension = 2; cfg.metric = DistanceMetric::Euclidean;
        VectorStore s(cfg);
        Chunk a = chunk_with("origin", {0.0f, 0.0f}), b = chunk_with("near", {1.0f, 0.0f}), c = chunk_with("far", {10.0f, 0.0f});
        ChunkId ia = a.id, ib = b.id;
        s.insert(a); s.insert(b); s.insert(c);
        auto r = s.search({0.0f, 0.0f}, 10);
        EXPECT(r[0].first == ia && r[1].first == ib);
        cfg.metric = DistanceMetric::DotProduct;
        VectorStore d(cfg);
        Chunk sm = chunk_with("small", {1.0f, 0.0f}), lg = chunk_with("large", {10.0f, 0.0f});
        ChunkId il = lg.id;
        d.insert(sm); d.insert(lg);
        EXPECT(d.search({1.0f, 0.0f}, 10)[0].first == il);
    }
    {   // RRF: src/fusion.rs:297-325
        FusionStrategy f;
        ChunkId a = ChunkId::from_u128(1), b = ChunkId::from_u128(2), c = ChunkId::from_u128(3);
        auto r = f.fuse({{a, 0.9f}, {b, 0.8f}}, {{a, 0.9f}, {c, 0.8f}});
        EXPECT(r.size() == 3 && r[0].first == a);
        auto one = f.fuse({{a, 1.0f}}, {{a, 1.0f}});
        EXPECT(std::fabs(one[0].second - 2.0f / 61.0f) < 1e-3f);
        EXPECT(f.fuse({}, {}).empty());
    }
    {   // hybrid + dense retrievers: shape checks as src/retrieve.rs:464-531
        HybridRetriever<StubEmbedder, StubSparse> hr(VectorStore::with_dimension(3), StubSparse{}, StubEmbedder{});
        std::vector<Chunk> cs = {chunk_with("alpha", {1, 0, 0}), chunk_with("beta", {0, 1, 0}), chunk_with("gamma", {0, 0, 1})};
        ChunkId first = cs[0].id;
        hr.index_batch(cs);
        EXPECT(hr.len() == 3);
        auto res = hr.retrieve("a", 2);
        EXPECT(res.size() == 2 && res[0].fused_score.has_value());
        EXPECT(res[0].best_score() >= res[1].best_score());
        auto dres = hr.retrieve_dense("a", 1);
        EXPECT(dres.size() == 1 && dres[0].chunk.id == first && dres[0].dense_score.has_value());
        DenseRetriever<StubEmbedder> dr(VectorStore::with_dimension(3), StubEmbedder{});
        dr.index(chunk_with("beta", {0, 1, 0}));
        EXPECT(dr.retrieve("b", 5).size() == 1);
    }
    {   // BM25Index: src/index.rs:481-665
        BM25Index ix;
        EXPECT(ix.len() == 0 && ix.is_empty() && std::fabs(ix.k1 - 1.2f) < 0.01f && std::fabs(ix.b - 0.75f) < 0.01f);
        BM25Index p = BM25Index::with_params(1.5f, 0.5f);
        EXPECT(std::fabs(p.k1 - 1.5f) < 0.01f && std::fabs(p.b - 0.5f) < 0.01f);
        auto toks = ix.tokenize("Hello World! This is a test.");
        auto has = [&](const char* w) { return std::find(toks.begin(), toks.end(), std::string(w)) != toks.end(); };
        EXPECT(has("hello") && has("world") && has("test") && !has("this") && !has("is") && !has("a"));
        Chunk c1("Machine learning algorithms"), c2("Deep learning neural networks"), c3("Natural language processing");
        ix.add(c1); ix.add(c2); ix.add(c3);
        EXPECT(ix.len() == 3 && ix.has_term("machine") && ix.has_term("learning"));
        auto r = ix.search("machine learning", 10);
        EXPECT(!r.empty() && r[0].first == c1.id);
        EXPECT(ix.search("", 10).empty() && ix.search("the a an", 10).empty() && ix.search("quantum physics", 10).empty());
        BM25Index rk;
        Chunk p1("python programming language"), p2("python python python programming");
        rk.add(p1); rk.add(p2);
        auto rr = rk.search("python programming", 10);
        EXPECT(rr.size() == 2 && rr[0].first == p2.id);
        BM25Index tk;
        for (int i = 0; i < 10; ++i) tk.add(Chunk("document " + std::to_string(i) + " about rust"));
        EXPECT(tk.search("rust", 3).size() == 3);
        BM25Index rm;
        Chunk t("Test document");
        rm.add(t);
        rm.remove(t.id);
        EXPECT(rm.len() == 0 && rm.search("test", 10).empty());
        // hybrid retriever with the real sparse index, src/retrieve.rs:175-220
        HybridRetriever<StubEmbedder, BM25Index> hr(VectorStore::with_dimension(3), BM25Index(), StubEmbedder{});
        std::vector<Chunk> cs = {chunk_with("alpha particles", {1, 0, 0}), chunk_with("beta decay", {0, 1, 0}),
                                 chunk_with("gamma rays and beta decay", {0, 0, 1})};
        hr.index_batch(cs);
        auto res = hr.retrieve("beta decay", 3);
        EXPECT(res.size() == 3 && res[0].chunk.id == cs[1].id && res[0].sparse_score.has_value());
    }
    {   // SparseRetriever: src/retrieve.rs:300-335, the reference's unit tests :568-658
        SparseRetriever<> empty;
        EXPECT(empty.retrieve("test", 10).empty());
        SparseRetriever<> one;
        one.index(Chunk("machine learning test"));
        auto r1 = one.retrieve("machine", 10);
        EXPECT(r1.size() == 1 && r1[0].sparse_score.has_value() && !r1[0].dense_score.has_value());
        auto two = SparseRetriever<BM25Index>::with_index(BM25Index::with_params(1.5f, 0.5f));
        Chunk a("rust programming language"), b("python programming language");
        const ChunkId ia = a.id;
        two.index(a);
        two.index(b);
        EXPECT(two.retrieve("programming", 10).size() == 2);
        EXPECT(two.retrieve("programming", 1)[0].chunk.id == ia);   // equal scores: insertion order
    }
    {   // insert_batch: one call for the new rows; a known id is overwritten, a repeated id keeps its last row
        VectorStore s = VectorStore::with_dimension(2);
        Chunk old = chunk_with("old", {1.0f, 0.0f});
        ChunkId oid = old.id;
        s.insert(old);
        Chunk again("old again");
        again.id = oid;
        again = std::move(again).with_embedding({0.0f, 1.0f});
        Chunk n1 = chunk_with("n1", {1.0f, 0.0f}), n2 = chunk_with("n2", {0.6f, 0.8f});
        Chunk n1b("n1 later");
        n1b.id = n1.id;
        n1b = std::move(n1b).with_embedding({-1.0f, 0.0f});
        ChunkId i1 = n1.id, i2 = n2.id;
        s.insert_batch({n1, again, n2, n1b});
        EXPECT(s.len() == 3);
        auto r = s.search({0.0f, 1.0f}, 1u << 30);       // any k: the reference truncates
        EXPECT(r.size() == 3 && r[0].first == oid && r[1].first == i2 && r[2].first == i1);
        EXPECT(s.get(i1)->content == "n1 later" && s.get(oid)->content == "old again");
        try { s.insert_batch({chunk_with("ok", {1.0f, 1.0f}), Chunk("no embedding")}); EXPECT(false); }
        catch (const InvalidConfig&) {}
        EXPECT(s.len() == 3);                            // nothing of a failed batch is inserted
    }
    {   // clone: same answers, independent afterwards (#[derive(Clone)], src/index.rs:322)
        VectorStore s = VectorStore::with_dimension(2);
        Chunk a = chunk_with("a", {1.0f, 0.0f}), b = chunk_with("b", {0.0f, 1.0f}), c = chunk_with("c", {0.6f, 0.8f});
        s.insert_batch({a, b, c});
        s.remove(b.id);
        VectorStore t = s.clone();
        EXPECT(t.len() == 2 && t.get(a.id) != nullptr && t.get(b.id) == nullptr);
        auto rs = s.search({0.0f, 1.0f}, 5), rt = t.search({0.0f, 1.0f}, 5);
        EXPECT(rs.size() == 2 && rt.size() == 2 && rs[0].first == c.id && rt[0].first == c.id && rs[0].second == rt[0].second);
        t.insert(chunk_with("d", {0.0f, 1.0f}));
        s.remove(c.id);
        EXPECT(s.len() == 1 && t.len() == 3 && t.search({0.0f, 1.0f}, 1)[0].second == 1.0f);
        EXPECT(s.search({0.0f, 1.0f}, 5).size() == 1);
    }
    {   // not in the reference: device-resident queries with deferred selects (trag_store_search_device on the
        // handle, mode 2): six queries enqueued back to back, one fence, the host entry's answers
        const size_t d = 64, n = 5000, k = 7, nq = 6;
        VectorStore s = VectorStore::with_dimension(d);
        std::vector<Chunk> cs;
        uint32_t x = 12345u;
        auto rnd = [&]() { x = x * 1664525u + 1013904223u; return (float)((x >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
        for (size_t i = 0; i < n; ++i) {
            std::vector<float> v(d);
            for (auto& e : v) e = rnd();
            cs.push_back(chunk_with("c" + std::to_string(i), v));
        }
        s.insert_batch(cs);
        std::vector<float> q(nq * d);
        for (auto& e : q) e = rnd();
        float* dq = nullptr; uint32_t* drows = nullptr; float* dscores = nullptr; uint32_t* dn = nullptr; uint32_t* dst = nullptr;
        EXPECT(hipMalloc((void**)&dq, q.size() * 4) == hipSuccess && hipMalloc((void**)&drows, nq * k * 4) == hipSuccess &&
               hipMalloc((void**)&dscores, nq * k * 4) == hipSuccess && hipMalloc((void**)&dn, nq * 4) == hipSuccess &&
               hipMalloc((void**)&dst, nq * 4) == hipSuccess);
        EXPECT(hipMemcpy(dq, q.data(), q.size() * 4, hipMemcpyHostToDevice) == hipSuccess);
        EXPECT(hipMemset(drows, 0xFF, nq * k * 4) == hipSuccess && hipMemset(dst, 0xFF, nq * 4) == hipSuccess);
        s.set_pipelined(2);
        for (size_t b = 0; b < nq; ++b)
            EXPECT(trag_store_search_device(s.handle(), dq + b * d, 1, (uint32_t)d, (uint32_t)k, drows + b * k, dscores + b * k,
                                            dn + b, dst + b, nullptr) == TRAG_OK);
        std::vector<uint32_t> early(k);
        EXPECT(hipDeviceSynchronize() == hipSuccess);
        EXPECT(hipMemcpy(early.data(), drows + (nq - 1) * k, k * 4, hipMemcpyDeviceToHost) == hipSuccess);
        EXPECT(early[0] == 0xFFFFFFFFu);                    // the last query's select has not even been launched
        s.fence(nullptr);
        EXPECT(hipDeviceSynchronize() == hipSuccess);
        std::vector<uint32_t> rows(nq * k), cnt(nq), st(nq);
        std::vector<float> scores(nq * k);
        EXPECT(hipMemcpy(rows.data(), drows, rows.size() * 4, hipMemcpyDeviceToHost) == hipSuccess);
        EXPECT(hipMemcpy(scores.data(), dscores, scores.size() * 4, hipMemcpyDeviceToHost) == hipSuccess);
        EXPECT(hipMemcpy(cnt.data(), dn, nq * 4, hipMemcpyDeviceToHost) == hipSuccess);
        EXPECT(hipMemcpy(st.data(), dst, nq * 4, hipMemcpyDeviceToHost) == hipSuccess);
        s.set_pipelined(0);
        for (size_t b = 0; b < nq; ++b) {
            auto want = s.search(std::vector<float>(q.begin() + b * d, q.begin() + (b + 1) * d), k);
            EXPECT(st[b] == 0 && cnt[b] == k && want.size() == k);
            for (size_t i = 0; i < k && i < want.size(); ++i)
                EXPECT(cs[rows[b * k + i]].id == want[i].first && scores[b * k + i] == want[i].second);
        }
        (void)hipFree(dq); (void)hipFree(drows); (void)hipFree(dscores); (void)hipFree(dn); (void)hipFree(dst);
    }
    {   // BM25: re-adding an id replaces its document; any k
        BM25Index ix;
        Chunk c("solar panels on the roof"), other("wind turbines at sea");
        ix.add(c); ix.add(other);
        Chunk c2("tidal power stations");
        c2.id = c.id;
        ix.add(c2);
        EXPECT(ix.len() == 2 && ix.search("solar", 10).empty());
        auto r = ix.search("tidal power", 1u << 30);
        EXPECT(r.size() == 1 && r[0].first == c.id);
    }
    std::printf(failures ? "%d failure(s)\n" : "host mirror: all checks passed\n", failures);
    return failures ? 1 : 0;
}
