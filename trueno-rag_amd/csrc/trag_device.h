// Device-side helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace trag {

typedef float v4f __attribute__((ext_vector_type(4)));

// Ordered 32-bit key of a score: larger key = better rank.
//   0            removed row (never a result)
//   1            NaN score (ranks after every number)
//   >= 0x007FFFFF  -inf .. +inf in numeric order; -0.0 and +0.0 share one key
// (the reference compares scores with partial_cmp, src/index.rs:408, for which
// the two zeros are Equal).
__device__ __forceinline__ uint32_t score_key(float s) {
    if (s != s) return 1u;
    s = s + 0.0f;  // -0.0 -> +0.0
    uint32_t b = __float_as_uint(s);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__device__ __forceinline__ float key_score(uint32_t k) {
    uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(b);  // key 1 -> 0xFFFFFFFE, a NaN
}

// 64-bit sort key: score key, then lower row first.
__device__ __forceinline__ unsigned long long rank_key(uint32_t skey, uint32_t row) {
    return ((unsigned long long)skey << 32) | (unsigned long long)(0xFFFFFFFFu - row);
}

__device__ __forceinline__ bool row_is_dead(const uint32_t* __restrict__ dead, uint32_t row) {
    return (dead[row >> 5] >> (row & 31)) & 1u;
}

// The row-dependent part of a score is v = fma(dot, scale, shift) -- cosine dot / ||v||, dot product
// dot, Euclidean 2 dot - ||v||^2: (scale, shift) of one row for the bf16 prefilter's passes. Rows
// that take no part (removed, past the end): scale 0, shift -inf. Rows whose norm is outside the
// bound's range must always be candidates: scale 0, shift +inf (the sample pass leaves them out).
// The reciprocal square root is the hardware's (1 ulp): v only has to be within the bound's 1 % slack
// of its value.
__device__ __forceinline__ float2 row_scale_shift(const float* __restrict__ norm2, const uint32_t* __restrict__ dead,
                                                  uint32_t n_rows, uint32_t metric, uint32_t row) {
    const bool live = row < n_rows && !row_is_dead(dead, row);
    const float nb2 = (metric == 2u || !live) ? 1.0f : norm2[row];
    const bool forced = !(nb2 == 0.0f || (nb2 >= 0x1p-100f && nb2 < INFINITY));
    float sc = metric == 0u ? (nb2 > 0.0f ? __builtin_amdgcn_rsqf(nb2) : 0.0f) : (metric == 1u ? 2.0f : 1.0f);
    float sh = metric == 1u ? -nb2 : 0.0f;
    sc = forced ? 0.0f : sc;
    sh = forced ? INFINITY : sh;
    sc = live ? sc : 0.0f;
    sh = live ? sh : -INFINITY;
    return make_float2(sc, sh);
}

// Largest value over the 64 lanes of a wave (every lane must call it).
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t w = (uint32_t)__shfl_xor((int)v, o);
        v = w > v ? w : v;
    }
    return v;
}

// find_bin over a 2048-bin histogram in LDS, called by all 256 threads of the workgroup: the highest
// bin whose suffix count reaches `need`, and the count strictly above it. sh: 6 words.
__device__ __forceinline__ void find_bin_256(const uint32_t* hist, uint32_t need, uint32_t& bin, uint32_t& above,
                                             uint32_t* sh) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int top = 2047 - 8 * t;  // thread t owns bins top, top - 1, ..., top - 7
    uint32_t h[8], sum = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        h[i] = hist[top - i];
        sum += h[i];
    }
    uint32_t incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) sh[w] = incl;
    if (t == 0) { sh[4] = 0; sh[5] = 0; }
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < w) base += sh[i];
        total += sh[i];
    }
    incl += base;
    if (need > total) need = total;
    uint32_t run = incl - sum;
    if (need > 0 && run < need && need <= incl) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (run < need && run + h[i] >= need) { sh[4] = (uint32_t)(top - i); sh[5] = run; }
            run += h[i];
        }
    }
    __syncthreads();
    bin = sh[4];
    above = sh[5];
    __syncthreads();
}

// bf16 prefilter: largest |approximate - reference| of a score, as a multiple of
// the scale ||q|| * ||v|| (cosine: of 1). With q' and v' the operands rounded to bf16,
//   |sum q'v' - sum qv| = |q'.(v' - v) + (q' - q).v| <= ||q'|| ||v' - v|| + ||q' - q|| ||v||
//                       <= (rho_q + rho_v + rho_q rho_v) ||q|| ||v||,
// rho_q = ||q' - q|| / ||q|| of this query and rho_v the largest ||v' - v|| / ||v|| over the
// corpus rows. Both are MEASURED, not assumed: the differences x' - x are exact in f32, the
// store keeps the largest squared ratio of its rows beside the largest squared norm (updated
// by every insert, never lowered) and the batch prep measures each query (bf16_operand_rho).
// bf16 keeps 8 significant bits, so the worst case is 2^-8 each (a vector whose every
// component sits just below a rounding boundary); embeddings measure near 0.4 * 2^-8, which
// makes the bound -- and with it the candidate lists -- 2.5 times tighter.
// Added to it: d f32 accumulation steps of unknown rounding inside the matrix cores
// (2^-22 each) and the rounding of the reference's own sequential sum and norms
// ((2d + 64) * 2^-24).
__device__ __forceinline__ float bf16_unit_eps(uint32_t d, float rho) {
    const float fd = (float)d;
    return (rho * 1.003f + fd * 2.3841858e-07f * 1.01f + (2.0f * fd + 64.0f) * 5.9604645e-08f) * 1.01f;
}

// rho of a query against a store. eq2 = sum (q' - q)^2 and na_sq = sum q^2 (any summation
// order: 1 % covers their rounding), rho_v_sq = the store's largest squared ratio (already
// inflated). Anything out of range: the format's worst case.
__device__ __forceinline__ float bf16_operand_rho(float eq2, float na_sq, float rho_v_sq) {
    const float worst = 2.0f * 0.00390625f + 1.52587890625e-05f;
    float rq = 0.0f;   // (a zero query has q' = q)
    if (na_sq > 0.0f) rq = sqrtf(eq2 / na_sq * 1.01f);
    const float rv = sqrtf(rho_v_sq);
    const float r = (rq + rv + rq * rv) * 1.001f;
    return (r >= 0.0f && r < worst) ? r : worst;
}

// sum over the components of (bf16(x) - x)^2, one component: the rounding is the matrix-core
// kernels' (and the mirror's) own conversion, the difference is exact
__device__ __forceinline__ float bf16_round_err_sq(float x, float acc) {
    const float r = (float)(__bf16)x - x;
    return __builtin_fmaf(r, r, acc);
}

// Single-query scan of the bf16 mirror: only the corpus is rounded (the query stays
// f32 and the products are accumulated in f32): 2^-8 ||q|| ||v|| from the rounding,
// plus the f32 summation error of both the approximate and the reference sum.
// METRIC 0 cosine and 2 dot product: in score space; 1 Euclidean: in score space too
// (| ||q - v'|| - ||q - v|| | <= ||v' - v|| <= 2^-8 ||v||).
template <int METRIC>
__device__ __forceinline__ float mirror_eps(uint32_t d, float na, float vmax) {
    const float r = 0.00390625f * 1.003f;
    // (cosine: 3d -- the two row norms are each off by d/2 * 2^-24 relative in either path)
    const float f = ((METRIC == 0 ? 3.0f : 2.0f) * (float)d + 64.0f) * 5.9604645e-08f * 1.01f;
    if (METRIC == 0) return (r + f) * 1.01f + 1e-30f;
    if (METRIC == 2) return (r + f) * na * vmax * 1.01f + 1e-30f;
    return (r * vmax + f * (na + vmax)) * 1.01f + 1e-20f;
}

// The same bound in the key space of each metric (0 cosine: score; 2 dot
// product: score; 1 Euclidean: negated squared distance).
template <int METRIC>
__device__ __forceinline__ float bf16_eps(uint32_t d, float na, float vmax, float rho) {
    const float c = bf16_unit_eps(d, rho);
    if (METRIC == 0) return c + 1e-30f;
    if (METRIC == 2) return c * na * vmax + 1e-30f;
    const float span = na + vmax;
    return 2.0f * c * na * vmax + (2.0f * (float)d + 64.0f) * 5.9604645e-08f * span * span * 1.01f + 1e-30f;
}

// bf16 prefilter, thresholds that tighten during the full pass. A query's candidate threshold
// (in the space of v, the row-dependent part of the score) starts at the value T0 its sample
// gave and may rise along the ladder  T(m) = fma(m, step, T0),  step = 2.02 eps / SUB:  once
// `need` different rows have been seen with v >= T(j + SUB) = T(j) + 2.02 eps, the need-th best
// approximate value A of the whole corpus is at least T(j) + 2.02 eps, every row of the exact
// top `need` has v >= A - 2 eps > T(j), and T(j) is as good a threshold as T0 was. Every party
// evaluates the ladder with this one expression, so a level means the same value everywhere.
__device__ __forceinline__ float dyn_level_value(uint32_t level, float t0, float step) {
    return __builtin_fmaf((float)level, step, t0);
}

// The highest level a row with approximate value v vouches for: the largest j with
// T(j + sub) <= v, at most levels - 1; negative: none (also for a v that is not finite -- rows
// outside the bound's range are candidates whatever the threshold and prove nothing -- and for a
// query without a ladder, step 0).
__device__ __forceinline__ int dyn_hit_level(float v, float t0, float step, int levels, int sub) {
    if (!(step > 0.0f) || !(v < INFINITY) || !(t0 > -INFINITY) || !(t0 < INFINITY)) return -1;
    float x = (v - t0) / step;
    if (!(x >= 0.0f)) return -1;
    x = fminf(x, (float)(levels + sub));
    int m = (int)x;
    // the ladder's own arithmetic has the last word (the quotient above may be one step off)
#pragma unroll
    for (int it = 0; it < 3; ++it)
        if (m >= 0 && !(dyn_level_value((uint32_t)m, t0, step) <= v)) --m;
    if (m < 0 || !(dyn_level_value((uint32_t)m, t0, step) <= v)) return -1;
    const int j = m - sub;
    return j < levels - 1 ? j : levels - 1;
}

// Same integer hash as the host generator used by the tests.
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ __forceinline__ float synth_component(unsigned long long seed, unsigned long long row,
                                                 uint32_t col) {
    unsigned long long h = splitmix64(seed ^ splitmix64(row * 0x100000001B3ull + col));
    int s = (int)(h & 0xFFFF) + (int)((h >> 16) & 0xFFFF) + (int)((h >> 32) & 0xFFFF) +
            (int)((h >> 48) & 0xFFFF) - 131070;
    return (float)s * (1.0f / 37837.0f);
}

}  // namespace trag
