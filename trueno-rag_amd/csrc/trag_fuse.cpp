// Host-side fusion of dense and sparse result lists (FusionStrategy::fuse,
// src/fusion.rs:42-231 of the reference) and the cross-shard top-k merge.
// Inputs are at most ~100 entries per call (src/retrieve.rs:94,176), so this
// is plain C++; it must be compiled without FMA contraction.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

#include "trag.h"
#include "trag_error.h"

namespace {

struct Entry {
    uint64_t id;
    float score;
    uint32_t order;  // first appearance (dense list, then sparse list) or Union rank
};

// Accumulating map that remembers first-appearance order: the deterministic
// stand-in for the reference's HashMap<ChunkId, f32>.
struct Accum {
    std::vector<Entry> items;
    std::unordered_map<uint64_t, uint32_t> at;
    float& slot(uint64_t id) {
        auto it = at.find(id);
        if (it == at.end()) {
            at.emplace(id, (uint32_t)items.size());
            items.push_back({id, 0.0f, (uint32_t)items.size()});
            return items.back().score;
        }
        return items[it->second].score;
    }
};

// score descending (NaN last), ties by `order` ascending. src/fusion.rs:227-231.
bool before(const Entry& a, const Entry& b) {
    const bool na = a.score != a.score, nb = b.score != b.score;
    if (na || nb) return na != nb ? nb : a.order < b.order;
    if (a.score > b.score) return true;
    if (a.score < b.score) return false;
    return a.order < b.order;
}

void emit(std::vector<Entry>& v, uint64_t* out_ids, float* out_scores, uint32_t* out_n) {
    for (size_t i = 0; i < v.size(); ++i) { out_ids[i] = v[i].id; out_scores[i] = v[i].score; }
    *out_n = (uint32_t)v.size();
}

// src/fusion.rs:183-202
std::vector<float> min_max(const float* s, uint32_t n) {
    std::vector<float> out(n);
    if (!n) return out;
    float mn = INFINITY, mx = -INFINITY;
    for (uint32_t i = 0; i < n; ++i) { mn = std::fmin(mn, s[i]); mx = std::fmax(mx, s[i]); }
    const float range = mx - mn;
    if (std::fabs(range) < 1.1920929e-07f) { std::fill(out.begin(), out.end(), 1.0f); return out; }
    for (uint32_t i = 0; i < n; ++i) out[i] = (s[i] - mn) / range;
    return out;
}

// src/fusion.rs:205-224
std::vector<float> z_score(const float* s, uint32_t n) {
    std::vector<float> out(n);
    if (!n) return out;
    const float fn = (float)n;
    float sum = 0.0f;
    for (uint32_t i = 0; i < n; ++i) sum = sum + s[i];
    const float mean = sum / fn;
    float var = 0.0f;
    for (uint32_t i = 0; i < n; ++i) { const float t = s[i] - mean; var = var + t * t; }
    var = var / fn;
    const float sd = std::sqrt(var);
    if (std::fabs(sd) < 1.1920929e-07f) { std::fill(out.begin(), out.end(), 0.0f); return out; }
    for (uint32_t i = 0; i < n; ++i) out[i] = (s[i] - mean) / sd;
    return out;
}

trag_status bad(const char* fn, const char* what) {
    return trag::set_error(TRAG_INVALID_ARG, std::string(fn) + ": " + what);
}

}  // namespace

extern "C" {

trag_status trag_fuse(uint32_t strategy, float param, const uint64_t* dense_ids,
                      const float* dense_scores, uint32_t nd, const uint64_t* sparse_ids,
                      const float* sparse_scores, uint32_t ns, uint64_t* out_ids, float* out_scores,
                      uint32_t* out_n) {
    if (!out_n) return bad("trag_fuse", "out_n is null");
    *out_n = 0;
    if ((nd && !dense_ids) || (ns && !sparse_ids)) return bad("trag_fuse", "a non-empty input list has no ids");
    if ((nd + ns) && (!out_ids || !out_scores)) return bad("trag_fuse", "null output array");
    if (strategy > 5) return bad("trag_fuse", "unknown strategy (0 RRF .. 5 Intersection)");
    if (strategy != 0 && ((nd && !dense_scores) || (ns && !sparse_scores)))
        return bad("trag_fuse", "this strategy reads the input scores, which are null");
    Accum acc;
    switch (strategy) {
    case 0: {  // RRF, src/fusion.rs:68-84: 1.0 / (k + rank as f32 + 1.0)
        for (uint32_t pass = 0; pass < 2; ++pass) {
            const uint64_t* ids = pass ? sparse_ids : dense_ids;
            const uint32_t n = pass ? ns : nd;
            for (uint32_t rank = 0; rank < n; ++rank) {
                float den = param + (float)rank;
                den = den + 1.0f;
                float& s = acc.slot(ids[rank]);
                s = s + 1.0f / den;
            }
        }
        break;
    }
    case 1:    // Linear, src/fusion.rs:87-109
    case 2: {  // Convex = Linear with alpha, src/fusion.rs:112-119
        const float wd = param, ws = 1.0f - param;
        const std::vector<float> dn = min_max(dense_scores, nd), sn = min_max(sparse_scores, ns);
        for (uint32_t i = 0; i < nd; ++i) { float& s = acc.slot(dense_ids[i]); s = s + wd * dn[i]; }
        for (uint32_t i = 0; i < ns; ++i) { float& s = acc.slot(sparse_ids[i]); s = s + ws * sn[i]; }
        break;
    }
    case 3: {  // DBSF, src/fusion.rs:122-138
        const std::vector<float> dn = z_score(dense_scores, nd), sn = z_score(sparse_scores, ns);
        for (uint32_t i = 0; i < nd; ++i) { float& s = acc.slot(dense_ids[i]); s = s + dn[i]; }
        for (uint32_t i = 0; i < ns; ++i) { float& s = acc.slot(sparse_ids[i]); s = s + sn[i]; }
        break;
    }
    case 4: {  // Union, src/fusion.rs:141-160: ordered by rank, not by score
        for (uint32_t i = 0; i < nd; ++i) {
            acc.slot(dense_ids[i]) = dense_scores[i];          // insert(): later duplicate overwrites
            acc.items[acc.at[dense_ids[i]]].order = i;
        }
        for (uint32_t i = 0; i < ns; ++i) {
            if (acc.at.find(sparse_ids[i]) != acc.at.end()) continue;  // or_insert
            acc.slot(sparse_ids[i]) = sparse_scores[i];
            acc.items.back().order = nd + i;
        }
        std::stable_sort(acc.items.begin(), acc.items.end(),
                         [](const Entry& a, const Entry& b) { return a.order < b.order; });
        emit(acc.items, out_ids, out_scores, out_n);
        return TRAG_OK;
    }
    case 5: {  // Intersection, src/fusion.rs:163-180: maps collect, last duplicate wins
        std::unordered_map<uint64_t, float> dm, sm;
        for (uint32_t i = 0; i < nd; ++i) dm[dense_ids[i]] = dense_scores[i];
        for (uint32_t i = 0; i < ns; ++i) sm[sparse_ids[i]] = sparse_scores[i];
        for (uint32_t i = 0; i < nd; ++i) {
            const uint64_t id = dense_ids[i];
            if (acc.at.find(id) != acc.at.end()) continue;
            auto it = sm.find(id);
            if (it == sm.end()) continue;
            acc.slot(id) = (dm[id] + it->second) / 2.0f;
        }
        break;
    }
    }
    std::stable_sort(acc.items.begin(), acc.items.end(), before);
    emit(acc.items, out_ids, out_scores, out_n);
    return TRAG_OK;
}

trag_status trag_fuse_rrf(const uint64_t* dense_ids, uint32_t nd, const uint64_t* sparse_ids,
                          uint32_t ns, float k, uint64_t* out_ids, float* out_scores,
                          uint32_t* out_n) {
    return trag_fuse(0, k, dense_ids, nullptr, nd, sparse_ids, nullptr, ns, out_ids, out_scores, out_n);
}

trag_status trag_merge_topk(const uint32_t* rows, const float* scores, const uint32_t* list_n,
                            uint32_t n_lists, uint32_t k_in, uint32_t k_out, uint32_t* out_rows,
                            float* out_scores, uint32_t* out_n) {
    if (!out_n) return bad("trag_merge_topk", "out_n is null");
    *out_n = 0;
    if (n_lists && (!rows || !scores || !list_n)) return bad("trag_merge_topk", "null input array");
    std::vector<Entry> all;
    for (uint32_t l = 0; l < n_lists; ++l) {
        const uint32_t n = std::min(list_n[l], k_in);
        for (uint32_t i = 0; i < n; ++i)
            all.push_back({rows[(size_t)l * k_in + i], scores[(size_t)l * k_in + i],
                           rows[(size_t)l * k_in + i]});
    }
    std::sort(all.begin(), all.end(), before);  // `order` = global row: a total order
    const uint32_t m = (uint32_t)std::min<size_t>(all.size(), k_out);
    if (m && (!out_rows || !out_scores)) return bad("trag_merge_topk", "null output array");
    for (uint32_t i = 0; i < m; ++i) { out_rows[i] = (uint32_t)all[i].id; out_scores[i] = all[i].score; }
    *out_n = m;
    return TRAG_OK;
}

}  // extern "C"
