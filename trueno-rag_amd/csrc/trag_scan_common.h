// Pieces shared by the corpus-scan kernels.
#pragma once
#include "trag_device.h"
#include "trag_internal.h"

namespace trag {

// sum q*q in the reference's order (src/index.rs:442), computed by one wave:
// lane l loads q[64c + l] (coalesced, two chunks ahead), the values are
// broadcast one at a time with v_readlane and added strictly left to right, so
// every lane ends with the same bits. All 64 lanes must call it.
__device__ __forceinline__ float query_norm_sq_wave(const float* __restrict__ q, uint32_t d, int lane) {
    const uint32_t n_chunks = (d + 63u) / 64u;
    auto load = [&](uint32_t c) -> float {
        const uint32_t j = c * 64u + (uint32_t)lane;
        return (c < n_chunks && j < d) ? q[j] : 0.0f;
    };
    float acc = 0.0f;
    float cur = load(0), n1 = load(1), n2 = load(2);
    for (uint32_t c = 0; c < n_chunks; ++c) {
        const float n3 = load(c + 3);
        const uint32_t len = min(64u, d - c * 64u);
        const int bits = __float_as_int(cur);
        if (len == 64u) {
#pragma unroll
            for (int i = 0; i < 64; ++i) {
                const float x = __int_as_float(__builtin_amdgcn_readlane(bits, i));
                const float p = x * x;
                acc = acc + p;
            }
        } else {
            for (uint32_t i = 0; i < len; ++i) {
                const float x = __int_as_float(__builtin_amdgcn_readlane(bits, (int)i));
                const float p = x * x;
                acc = acc + p;
            }
        }
        cur = n1; n1 = n2; n2 = n3;
    }
    return acc;
}

// First workgroup of a scan publishes the per-query constants the later stages read.
__device__ __forceinline__ void publish_ctl(SearchCtl* __restrict__ ctl, float na_sq, uint32_t need,
                                            uint32_t k, const float* __restrict__ vmax_sq,
                                            uint32_t n_waves) {
    ctl->n_waves = n_waves;  // tile t is scanned by wave t % n_waves
    ctl->na_sq = na_sq;
    ctl->na = sqrtf(na_sq);
    ctl->vmax_sq = vmax_sq ? *vmax_sq : 0.0f;
    ctl->need = need;
    ctl->k = k;
    ctl->rho = 0.0f;   // (only the bf16 prefilter's batch prep measures it)
}

// ---------------------------------------------------------------------------
// Sparse key stores.
//
// A per-row key store costs a streaming scan far more than its bytes: on MI355X the
// mirror scan's loads and arithmetic stream at 7.1 TB/s, and one 256-byte key store per
// 48 KB tile (0.5 % of the bytes) takes 8 % off that; the cost follows the number of tiles
// stored, whatever the store policy or batching. The select stage only visits tiles whose
// maximum reaches its threshold, so a scan wave may skip the key store of a tile that no
// later stage can visit.
//
// The bound: every wave publishes the best (bounded) key it has seen so far in wbest[wave],
// tagged with the scan's epoch, after its 1st, 4th, 16th and 64th tile; at the start of its
// 3rd, 9th, 33rd and 129th tile the first wave of each workgroup reads all entries (the loads
// are issued there and used when the tile is done), takes their count-th largest value V and
// leaves the bound in LDS for the workgroup's other waves. The entries are keys of distinct
// rows, so at least `count` rows have a key >= V, the count-th best key of the corpus is >= V,
// and no stage needs a row below V (lowered by the metric-space margin where the select lowers
// its threshold): tiles whose maximum is below that need no keys. Entries only grow, so the value computed from the entries left at the
// end of the scan is >= every value a wave used: the select computes it with the same
// function and raises its cut to it, and thereby never visits a tile that was skipped.
// Entries of other epochs read as 0 (nothing is ever reset).
// ---------------------------------------------------------------------------
// count-th largest of the 1024 values held 16 per lane, rounded down to a multiple of 4096
// (0 when fewer than `count` reach 4096); every lane of the wave calls it and gets the same
// result. A bitwise search from the top; the counting is one ballot per register.
__device__ __forceinline__ uint32_t wave_kth_largest_1024(const uint32_t (&v)[16], uint32_t count) {
    uint32_t res = 0u;
    for (int bit = 31; bit >= 12; --bit) {
        const uint32_t cand = res | (1u << bit);
        uint32_t c = 0u;
#pragma unroll
        for (int i = 0; i < 16; ++i) c += (uint32_t)__popcll(__ballot(v[i] >= cand));
        if (c >= count) res = cand;
    }
    return res;
}

__device__ __forceinline__ void sparse_poll_issue(const unsigned long long* __restrict__ wbest, int lane,
                                                  unsigned long long (&e)[16]) {
#pragma unroll
    for (int i = 0; i < 16; ++i)   // (which lane holds which entry does not matter: coalesced)
        e[i] = __hip_atomic_load(wbest + i * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint32_t sparse_poll_value(const unsigned long long (&e)[16], uint32_t epoch,
                                                      uint32_t count) {
    uint32_t v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = (uint32_t)(e[i] >> 32) == epoch ? (uint32_t)e[i] : 0u;
    return wave_kth_largest_1024(v, count);
}

// the key below which no stage looks, from the count-th largest published key
__device__ __forceinline__ uint32_t sparse_keep_key(uint32_t kth, float margin) {
    if (kth < 0x007FFFFFu) return 0u;
    const float sc = key_score(kth) - margin;
    return (sc == sc && sc > -INFINITY) ? score_key(sc) : 0u;
}

// tiles after which a wave publishes / before which it polls (1-based tile counts)
__device__ __forceinline__ bool sparse_publish_at(uint32_t done) { return done == 1u || done == 4u || done == 16u || done == 64u; }
__device__ __forceinline__ bool sparse_poll_at(uint32_t done) { return done == 2u || done == 8u || done == 32u || done == 128u; }

}  // namespace trag
