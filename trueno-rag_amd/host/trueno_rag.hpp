// C++ host mirror of the part of trueno-rag's API that sits on the dense
// retrieval path, over the C ABI of include/trag.h. Names, argument meaning
// and error behaviour follow the reference (paths relative to its tree):
//   VectorStore, VectorStoreConfig, DistanceMetric     src/index.rs:282-437
//   BM25Index (SparseIndex)                             src/index.rs:7-280
//   FusionStrategy                                      src/fusion.rs:9-63
//   RetrievalResult, HybridRetrieverConfig,
//   HybridRetriever, DenseRetriever                     src/retrieve.rs:13-297
// Result<T, Error> becomes exceptions derived from trueno_rag::Error.
// Header-only; link with -ltrag. There is no CPU fallback behind it.
#pragma once
#include <algorithm>
#include <cstdint>
#include <optional>
#include <random>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

#include "trag.h"

namespace trueno_rag {

// ---- errors (src/error.rs:9-64) ----
struct Error : std::runtime_error { using std::runtime_error::runtime_error; };
struct DimensionMismatch : Error {
    size_t expected, actual;
    DimensionMismatch(size_t e, size_t a)
        : Error("dimension mismatch: expected " + std::to_string(e) + ", got " + std::to_string(a)),
          expected(e), actual(a) {}
};
struct InvalidConfig : Error { using Error::Error; };
struct VectorStoreError : Error { using Error::Error; };  // Error::VectorStore(String)
struct QueryError : Error { using Error::Error; };        // Error::Query(String)

inline void check(int status, size_t expected = 0, size_t actual = 0) {
    if (status == TRAG_OK) return;
    if (status == TRAG_DIM_MISMATCH) throw DimensionMismatch(expected, actual);
    if (status == TRAG_INVALID_CONFIG) throw InvalidConfig(trag_last_error());
    throw VectorStoreError(trag_last_error());
}

// ---- ids and chunks (src/chunk.rs:7-99) ----
struct ChunkId {
    uint64_t hi = 0, lo = 0;
    static ChunkId random() {  // Uuid::new_v4
        static thread_local std::mt19937_64 g{std::random_device{}()};
        return ChunkId{g(), g()};
    }
    static ChunkId from_u128(uint64_t lo) { return ChunkId{0, lo}; }
    bool operator==(const ChunkId& o) const { return hi == o.hi && lo == o.lo; }
    bool operator!=(const ChunkId& o) const { return !(*this == o); }
};
struct ChunkIdHash {
    size_t operator()(const ChunkId& c) const { return std::hash<uint64_t>()(c.hi * 0x9E3779B97F4A7C15ull ^ c.lo); }
};

struct Chunk {
    ChunkId id = ChunkId::random();
    std::string content;
    std::optional<std::vector<float>> embedding;
    Chunk() = default;
    explicit Chunk(std::string text) : content(std::move(text)) {}
    Chunk with_embedding(std::vector<float> e) && { embedding = std::move(e); return std::move(*this); }
};

// ---- VectorStore ----
enum class DistanceMetric : uint32_t { Cosine = 0, Euclidean = 1, DotProduct = 2 };

struct VectorStoreConfig {
    size_t dimension = 384;
    DistanceMetric metric = DistanceMetric::Cosine;
    size_t hnsw_m = 16, hnsw_ef_construction = 100, hnsw_ef_search = 50;  // never read, as in the reference
    int device = -1;          // not in the reference: HIP device of this store
};

class VectorStore {
  public:
    explicit VectorStore(VectorStoreConfig config) : config_(config) {
        trag_config c{(uint32_t)config.dimension, (uint32_t)config.metric, config.device, TRAG_SCAN_AUTO, 0};
        check(trag_store_create(&c, &h_));
    }
    static VectorStore with_dimension(size_t dimension) {
        VectorStoreConfig c;
        c.dimension = dimension;
        return VectorStore(c);
    }
    VectorStore(VectorStore&& o) noexcept { *this = std::move(o); }
    VectorStore& operator=(VectorStore&& o) noexcept {
        if (this != &o) {
            if (h_) trag_store_destroy(h_);
            h_ = o.h_; o.h_ = nullptr;
            config_ = o.config_; row_of_ = std::move(o.row_of_); id_of_ = std::move(o.id_of_);
            chunks_ = std::move(o.chunks_);
        }
        return *this;
    }
    VectorStore(const VectorStore&) = delete;
    VectorStore& operator=(const VectorStore&) = delete;
    ~VectorStore() { if (h_) trag_store_destroy(h_); }

    const VectorStoreConfig& config() const { return config_; }

    // #[derive(Clone)], src/index.rs:322: an independent store with the same rows (copied inside
    // HBM), ids and chunks. (Explicit, not a copy constructor: a copy is gigabytes.)
    VectorStore clone() const {
        trag_store* h = nullptr;
        check(trag_store_clone(h_, &h));
        VectorStore other(config_, h);
        other.row_of_ = row_of_;
        other.id_of_ = id_of_;
        other.chunks_ = chunks_;
        return other;
    }

    // src/index.rs:359-375
    void insert(Chunk chunk) {
        if (!chunk.embedding) throw InvalidConfig("chunk must have embedding");
        const std::vector<float>& e = *chunk.embedding;
        if (e.size() != config_.dimension) throw DimensionMismatch(config_.dimension, e.size());
        auto it = row_of_.find(chunk.id);
        if (it != row_of_.end()) {  // HashMap::insert overwrites
            check(trag_store_overwrite(h_, it->second, e.data(), (uint32_t)e.size()), config_.dimension, e.size());
        } else {
            uint32_t first = 0;
            check(trag_store_insert(h_, e.data(), 1, (uint32_t)e.size(), &first), config_.dimension, e.size());
            row_of_[chunk.id] = first;
            if (id_of_.size() <= first) id_of_.resize(first + 1);
            id_of_[first] = chunk.id;
        }
        const ChunkId id = chunk.id;
        chunks_.insert_or_assign(id, std::move(chunk));
    }
    // src/index.rs:378-383
    // One call into the library for the new rows of the batch (ids already present are overwritten
    // in place; an id that occurs twice in the batch keeps its last embedding). A chunk without an
    // embedding or of the wrong dimension fails the call before anything is inserted.
    void insert_batch(std::vector<Chunk> chunks) {
        for (const auto& c : chunks) {
            if (!c.embedding) throw InvalidConfig("chunk must have embedding");
            if (c.embedding->size() != config_.dimension) throw DimensionMismatch(config_.dimension, c.embedding->size());
        }
        std::vector<Chunk> fresh;
        std::unordered_map<ChunkId, size_t, ChunkIdHash> slot;
        for (auto& c : chunks) {
            if (row_of_.count(c.id)) { insert(std::move(c)); continue; }
            auto it = slot.find(c.id);
            if (it != slot.end()) { fresh[it->second] = std::move(c); continue; }
            slot.emplace(c.id, fresh.size());
            fresh.push_back(std::move(c));
        }
        if (fresh.empty()) return;
        std::vector<float> flat;
        flat.reserve(fresh.size() * config_.dimension);
        for (const auto& c : fresh) flat.insert(flat.end(), c.embedding->begin(), c.embedding->end());
        uint32_t first = 0;
        check(trag_store_insert(h_, flat.data(), fresh.size(), (uint32_t)config_.dimension, &first), config_.dimension,
              config_.dimension);
        if (id_of_.size() < (size_t)first + fresh.size()) id_of_.resize((size_t)first + fresh.size());
        for (size_t i = 0; i < fresh.size(); ++i) {
            const ChunkId id = fresh[i].id;
            row_of_[id] = first + (uint32_t)i;
            id_of_[first + i] = id;
            chunks_.insert_or_assign(id, std::move(fresh[i]));
        }
    }
    // src/index.rs:386-412
    std::vector<std::pair<ChunkId, float>> search(const std::vector<float>& query_vector, size_t k) const {
        if (query_vector.size() != config_.dimension) throw DimensionMismatch(config_.dimension, query_vector.size());
        k = std::min(k, len());  // the reference truncates: any k is legal
        std::vector<uint32_t> rows(k ? k : 1);
        std::vector<float> scores(k ? k : 1);
        uint32_t n = 0;
        check(trag_store_search(h_, query_vector.data(), (uint32_t)query_vector.size(), (uint32_t)k, rows.data(),
                                scores.data(), &n),
              config_.dimension, query_vector.size());
        std::vector<std::pair<ChunkId, float>> out;
        out.reserve(n);
        for (uint32_t i = 0; i < n; ++i) out.emplace_back(id_of_[rows[i]], scores[i]);
        return out;
    }
    // src/index.rs:416-418
    const Chunk* get(ChunkId id) const {
        auto it = chunks_.find(id);
        return it == chunks_.end() ? nullptr : &it->second;
    }
    // src/index.rs:421-424
    std::optional<Chunk> remove(ChunkId id) {
        auto it = row_of_.find(id);
        if (it == row_of_.end()) return std::nullopt;
        check(trag_store_remove(h_, it->second));
        row_of_.erase(it);
        auto c = chunks_.find(id);
        std::optional<Chunk> out;
        if (c != chunks_.end()) { out = std::move(c->second); chunks_.erase(c); }
        return out;
    }
    size_t len() const { return (size_t)trag_store_len(h_); }
    bool is_empty() const { return len() == 0; }
    // Not in the reference: keep a bf16 copy of the corpus in HBM (1.5 x the memory) that
    // prefiltering passes read instead of the f32 rows; results stay the exact f32 results.
    void set_bf16_mirror(bool enabled) { check(trag_store_set_bf16_mirror(h_, enabled ? 1 : 0)); }
    bool has_bf16_mirror() const { return trag_store_has_bf16_mirror(h_) != 0; }
    // Not in the reference: callers whose queries and results live in device memory use trag_store_search_device on
    // handle(); these two go with it. Mode 2 defers the select of a single query into the launch of the next
    // query's scan, mode 1 runs it on a side stream, 0 switches both off; results (and, in mode 2, even the
    // launch of the last query's select) are behind fence(stream).
    void set_pipelined(int mode) { check(trag_store_set_pipelined(h_, mode)); }
    void fence(void* hip_stream = nullptr) const { check(trag_store_fence(h_, hip_stream)); }
    trag_store* handle() const { return h_; }

  private:
    VectorStore(VectorStoreConfig config, trag_store* adopted) : config_(config), h_(adopted) {}
    VectorStoreConfig config_;
    trag_store* h_ = nullptr;
    std::unordered_map<ChunkId, uint32_t, ChunkIdHash> row_of_;
    std::vector<ChunkId> id_of_;
    std::unordered_map<ChunkId, Chunk, ChunkIdHash> chunks_;
};

// ---- BM25Index (src/index.rs:31-280) ----
// Tokenizer and term dictionary on the host, scoring and top-k on the GPU
// (trag_bm25_*). The tokenizer here classifies ASCII letters and digits as
// alphanumeric and treats every byte >= 0x80 as part of a token (so UTF-8
// words stay whole); lower-casing is ASCII only. The reference uses Unicode
// char::is_alphanumeric / to_lowercase (src/index.rs:112-116): identical on
// ASCII text.
class BM25Index {
  public:
    BM25Index() : BM25Index(1.2f, 0.75f) {}  // BM25Index::new, :62-74
    static BM25Index with_params(float k1, float b) { return BM25Index(k1, b); }  // :78-84
    BM25Index(float k1_, float b_) : k1(k1_), b(b_), stopwords_(default_stopwords()) {
        trag_bm25_config c{k1_, b_, -1};
        check(trag_bm25_create(&c, &h_));
    }
    BM25Index(BM25Index&& o) noexcept
        : k1(o.k1), b(o.b), h_(o.h_), stopwords_(std::move(o.stopwords_)), terms_(std::move(o.terms_)),
          docs_of_(std::move(o.docs_of_)), id_of_(std::move(o.id_of_)) { o.h_ = nullptr; }
    BM25Index(const BM25Index&) = delete;
    BM25Index& operator=(const BM25Index&) = delete;
    ~BM25Index() { if (h_) trag_bm25_destroy(h_); }

    BM25Index with_stopwords(std::unordered_set<std::string> stopwords) && {  // :88-91
        stopwords_ = std::move(stopwords);
        return std::move(*this);
    }

    std::vector<std::string> tokenize(const std::string& text) const {  // :111-124
        std::vector<std::string> out;
        std::string cur;
        auto flush = [&]() {
            if (!cur.empty() && !stopwords_.count(cur) && cur.size() >= 2) out.push_back(cur);
            cur.clear();
        };
        for (unsigned char ch : text) {
            const bool alnum = (ch >= '0' && ch <= '9') || (ch >= 'a' && ch <= 'z') || (ch >= 'A' && ch <= 'Z') || ch >= 0x80;
            if (alnum) cur.push_back((ch >= 'A' && ch <= 'Z') ? (char)(ch + 32) : (char)ch);
            else flush();
        }
        flush();
        return out;
    }

    // Re-adding a ChunkId REPLACES its document. (The reference keeps both postings, counts the chunk
    // twice in doc_count and df while doc_lengths holds one entry, :176-204; that bookkeeping is not
    // reproduced: a chunk is one document here, so a search never returns an id twice and fuse never
    // counts it twice.)
    void add(const Chunk& chunk) {  // :176-204
        remove(chunk.id);
        const std::vector<uint32_t> ids = term_ids(tokenize(chunk.content), true);
        const uint64_t off[2] = {0, ids.size()};
        const uint32_t dummy = 0;
        uint32_t first = 0;
        check(trag_bm25_add(h_, ids.empty() ? &dummy : ids.data(), off, 1, &first));
        docs_of_[chunk.id].push_back(first);
        if (id_of_.size() <= first) id_of_.resize((size_t)first + 1);
        id_of_[first] = chunk.id;
    }
    void add_batch(const std::vector<Chunk>& chunks) { for (const auto& c : chunks) add(c); }  // :206-210

    std::vector<std::pair<ChunkId, float>> search(const std::string& query, size_t k) const {  // :212-243
        const std::vector<std::string> toks = tokenize(query);
        std::vector<std::pair<ChunkId, float>> out;
        // any k is legal in the reference (it truncates); the GPU select serves up to 4096 results
        k = std::min<size_t>(std::min(k, len()), 4096);
        if (toks.empty() || k == 0) return out;
        const std::vector<uint32_t> ids = const_cast<BM25Index*>(this)->term_ids(toks, false);
        std::vector<uint32_t> docs(k);
        std::vector<float> scores(k);
        uint32_t n = 0;
        check(trag_bm25_search(h_, ids.data(), (uint32_t)ids.size(), (uint32_t)k, docs.data(), scores.data(), &n));
        for (uint32_t i = 0; i < n; ++i) out.emplace_back(id_of_[docs[i]], scores[i]);
        return out;
    }

    void remove(ChunkId chunk_id) {  // :245-275
        auto it = docs_of_.find(chunk_id);
        if (it == docs_of_.end()) return;
        for (uint32_t d : it->second) check(trag_bm25_remove(h_, d));
        docs_of_.erase(it);
    }
    size_t len() const { return (size_t)trag_bm25_len(h_); }  // :277-279
    bool is_empty() const { return len() == 0; }               // :25-27
    bool has_term(const std::string& t) const { return terms_.count(t) != 0; }

    float k1, b;

  private:
    static std::unordered_set<std::string> default_stopwords() {  // :93-108
        static const char* w[] = {
            "a", "an", "the", "is", "are", "was", "were", "be", "been", "being", "have", "has", "had", "do", "does",
            "did", "will", "would", "could", "should", "may", "might", "must", "shall", "can", "need", "dare", "ought",
            "used", "to", "of", "in", "for", "on", "with", "at", "by", "from", "as", "into", "through", "during",
            "before", "after", "above", "below", "between", "under", "again", "further", "then", "once", "here",
            "there", "when", "where", "why", "how", "all", "each", "few", "more", "most", "other", "some", "such", "no",
            "nor", "not", "only", "own", "same", "so", "than", "too", "very", "just", "and", "but", "if", "or",
            "because", "until", "while", "this", "that", "these", "those", "it", "its"};
        return std::unordered_set<std::string>(std::begin(w), std::end(w));
    }
    std::vector<uint32_t> term_ids(const std::vector<std::string>& toks, bool create) {
        std::vector<uint32_t> ids;
        ids.reserve(toks.size());
        for (const auto& t : toks) {
            auto it = terms_.find(t);
            if (it != terms_.end()) ids.push_back(it->second);
            else if (create) { const uint32_t id = (uint32_t)terms_.size(); terms_.emplace(t, id); ids.push_back(id); }
            else ids.push_back(0x0FFFFFFFu);  // never assigned: matches nothing
        }
        return ids;
    }
    trag_bm25* h_ = nullptr;
    std::unordered_set<std::string> stopwords_;
    std::unordered_map<std::string, uint32_t> terms_;
    std::unordered_map<ChunkId, std::vector<uint32_t>, ChunkIdHash> docs_of_;
    std::vector<ChunkId> id_of_;
};

// ---- FusionStrategy (src/fusion.rs:9-63) ----
struct FusionStrategy {
    enum Kind : uint32_t { RRF = 0, Linear = 1, Convex = 2, DBSF = 3, Union = 4, Intersection = 5 };
    Kind kind = RRF;
    float param = 60.0f;  // k / dense_weight / alpha; Default = RRF{k: 60}, :33-37
    static FusionStrategy rrf(float k = 60.0f) { return {RRF, k}; }
    static FusionStrategy linear(float dense_weight) { return {Linear, dense_weight}; }
    static FusionStrategy convex(float alpha) { return {Convex, alpha}; }
    static FusionStrategy dbsf() { return {DBSF, 0.0f}; }
    static FusionStrategy union_() { return {Union, 0.0f}; }
    static FusionStrategy intersection() { return {Intersection, 0.0f}; }

    using Scored = std::vector<std::pair<ChunkId, float>>;
    Scored fuse(const Scored& dense_results, const Scored& sparse_results) const {
        std::unordered_map<ChunkId, uint64_t, ChunkIdHash> tok;
        std::vector<ChunkId> back;
        auto token = [&](const ChunkId& id) {
            auto it = tok.find(id);
            if (it != tok.end()) return it->second;
            tok.emplace(id, back.size());
            back.push_back(id);
            return (uint64_t)back.size() - 1;
        };
        std::vector<uint64_t> di, si;
        std::vector<float> ds, ss;
        for (auto& p : dense_results) { di.push_back(token(p.first)); ds.push_back(p.second); }
        for (auto& p : sparse_results) { si.push_back(token(p.first)); ss.push_back(p.second); }
        std::vector<uint64_t> oi(di.size() + si.size() + 1);
        std::vector<float> os(oi.size());
        uint32_t n = 0;
        check(trag_fuse(kind, param, di.data(), ds.data(), (uint32_t)di.size(), si.data(), ss.data(),
                        (uint32_t)si.size(), oi.data(), os.data(), &n));
        Scored out;
        for (uint32_t i = 0; i < n; ++i) out.emplace_back(back[oi[i]], os[i]);
        return out;
    }
};

// ---- retrievers (src/retrieve.rs) ----
struct RetrievalResult {
    Chunk chunk;
    std::optional<float> dense_score, sparse_score, fused_score, rerank_score;
    float best_score() const {  // :69-75
        if (rerank_score) return *rerank_score;
        if (fused_score) return *fused_score;
        if (dense_score) return *dense_score;
        if (sparse_score) return *sparse_score;
        return 0.0f;
    }
};

struct HybridRetrieverConfig {  // :80-100
    size_t candidates_per_source = 50;
    FusionStrategy fusion{};
    bool use_dense = true, use_sparse = true;
};

// E: embed_query(const std::string&) -> std::vector<float>   (Embedder, src/embed.rs:55-89)
// S: add(const Chunk&), search(const std::string&, size_t) -> vector<pair<ChunkId,float>>  (SparseIndex)
template <class E, class S>
class HybridRetriever {
  public:
    HybridRetriever(VectorStore dense, S sparse, E embedder)
        : dense_(std::move(dense)), sparse_(std::move(sparse)), embedder_(std::move(embedder)) {}
    HybridRetriever& with_config(HybridRetrieverConfig c) { config_ = c; return *this; }
    VectorStore& dense_store() { return dense_; }
    S& sparse_index() { return sparse_; }
    void index(Chunk chunk) {  // :156-164
        sparse_.add(chunk);
        dense_.insert(std::move(chunk));
    }
    void index_batch(std::vector<Chunk> chunks) { for (auto& c : chunks) index(std::move(c)); }
    std::vector<RetrievalResult> retrieve(const std::string& query, size_t k) {  // :175-220
        FusionStrategy::Scored dense_results, sparse_results;
        if (config_.use_dense) dense_results = dense_.search(embedder_.embed_query(query), config_.candidates_per_source);
        if (config_.use_sparse) sparse_results = sparse_.search(query, config_.candidates_per_source);
        auto fused = config_.fusion.fuse(dense_results, sparse_results);
        std::unordered_map<ChunkId, float, ChunkIdHash> dsc(dense_results.begin(), dense_results.end());
        std::unordered_map<ChunkId, float, ChunkIdHash> ssc(sparse_results.begin(), sparse_results.end());
        std::vector<RetrievalResult> out;
        for (size_t i = 0; i < fused.size() && i < k; ++i) {
            const Chunk* c = dense_.get(fused[i].first);
            if (!c) continue;  // ids unknown to the dense store are dropped (:205)
            RetrievalResult r{*c, {}, {}, fused[i].second, {}};
            if (auto it = dsc.find(fused[i].first); it != dsc.end()) r.dense_score = it->second;
            if (auto it = ssc.find(fused[i].first); it != ssc.end()) r.sparse_score = it->second;
            out.push_back(std::move(r));
        }
        return out;
    }
    std::vector<RetrievalResult> retrieve_dense(const std::string& query, size_t k) {  // :223-235
        std::vector<RetrievalResult> out;
        for (auto& p : dense_.search(embedder_.embed_query(query), k))
            if (const Chunk* c = dense_.get(p.first)) out.push_back(RetrievalResult{*c, p.second, {}, {}, {}});
        return out;
    }
    size_t len() const { return dense_.len(); }
    bool is_empty() const { return dense_.is_empty(); }

  private:
    VectorStore dense_;
    S sparse_;
    E embedder_;
    HybridRetrieverConfig config_{};
};

template <class E>
class DenseRetriever {  // :266-297
  public:
    DenseRetriever(VectorStore store, E embedder) : store_(std::move(store)), embedder_(std::move(embedder)) {}
    void index(Chunk chunk) { store_.insert(std::move(chunk)); }
    std::vector<RetrievalResult> retrieve(const std::string& query, size_t k) {
        std::vector<RetrievalResult> out;
        for (auto& p : store_.search(embedder_.embed_query(query), k))
            if (const Chunk* c = store_.get(p.first)) out.push_back(RetrievalResult{*c, p.second, {}, {}, {}});
        return out;
    }

  private:
    VectorStore store_;
    E embedder_;
};

// src/retrieve.rs:300-335: sparse-only retrieval over any SparseIndex S (add, search); the chunks are kept
// here, keyed by id, as the reference keeps them.
template <class S = BM25Index>
class SparseRetriever {
  public:
    SparseRetriever() = default;
    explicit SparseRetriever(S index) : index_(std::move(index)) {}
    static SparseRetriever with_index(S index) { return SparseRetriever(std::move(index)); }
    void index(Chunk chunk) {  // :316-319
        index_.add(chunk);
        const ChunkId id = chunk.id;
        chunks_.insert_or_assign(id, std::move(chunk));
    }
    std::vector<RetrievalResult> retrieve(const std::string& query, size_t k) {  // :323-334
        std::vector<RetrievalResult> out;
        for (auto& p : index_.search(query, k))
            if (auto it = chunks_.find(p.first); it != chunks_.end())
                out.push_back(RetrievalResult{it->second, {}, p.second, {}, {}});
        return out;
    }

  private:
    S index_{};
    std::unordered_map<ChunkId, Chunk, ChunkIdHash> chunks_;
};

}  // namespace trueno_rag
