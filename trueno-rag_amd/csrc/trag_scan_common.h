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
}

}  // namespace trag
