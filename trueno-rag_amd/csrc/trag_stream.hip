// Streaming corpus scan for gfx950: the bandwidth-bound single-query pass.
//
// Rows are read exactly once with 16-byte non-temporal loads issued along the
// row (a group of G lanes covers G*16 contiguous bytes per load), the query
// sits in registers (staged once per workgroup through LDS), partial sums are
// combined with a wavefront butterfly, and each 64-row tile ends with one
// coalesced store of 64 ordered keys plus 64 LDS histogram increments. The
// sums are re-associated (lane-strided FMA chains + butterfly), so the scores
// written here are approximations of the reference's sequential sums; the
// final stage re-scores the selected candidates exactly and certifies the
// result against a rigorous bound on |approximate - exact|.
#include "trag_device.h"
#include "trag_internal.h"
#include "trag_scan_common.h"

namespace trag {

template <int METRIC>
__device__ __forceinline__ void accum_fast(const v4f q, const v4f x, float& a0, float& a1) {
    if (METRIC == 1) {
        const v4f t = q - x;
        a0 = __builtin_fmaf(t.x, t.x, a0);
        a0 = __builtin_fmaf(t.y, t.y, a0);
        a0 = __builtin_fmaf(t.z, t.z, a0);
        a0 = __builtin_fmaf(t.w, t.w, a0);
    } else {
        a0 = __builtin_fmaf(q.x, x.x, a0);
        a0 = __builtin_fmaf(q.y, x.y, a0);
        a0 = __builtin_fmaf(q.z, x.z, a0);
        a0 = __builtin_fmaf(q.w, x.w, a0);
        if (METRIC == 0) {
            a1 = __builtin_fmaf(x.x, x.x, a1);
            a1 = __builtin_fmaf(x.y, x.y, a1);
            a1 = __builtin_fmaf(x.z, x.z, a1);
            a1 = __builtin_fmaf(x.w, x.w, a1);
        }
    }
}

// Rows whose approximate sums left the range in which the error bound of the
// certification holds (overflow, NaN, squared norm in the subnormal range) get
// +inf: they always reach the exact re-score, which decides their true rank.
template <int METRIC>
__device__ __forceinline__ float finish_fast(float a0, float a1, float na) {
    bool ok = fabsf(a0) < INFINITY;
    if (METRIC == 0) ok = ok && (a1 == 0.0f || (a1 >= 0x1p-100f && a1 < INFINITY));
    if (!ok) return INFINITY;
    if (METRIC == 0) {
        const float nb = sqrtf(a1);
        if (na == 0.0f || nb == 0.0f) return 0.0f;
        return a0 / (na * nb);
    } else if (METRIC == 1) {
        return -sqrtf(a0);
    }
    return a0;
}

// G lanes per row (32 or 64), NV float4 per lane per row, U rows-iterations in flight.
template <int METRIC, int G, int NV, int U>
__global__ void __launch_bounds__(256)
scan_stream_kernel(const float* __restrict__ corpus, const float* __restrict__ q,
                   const uint32_t* __restrict__ dead, uint32_t n_rows, uint32_t need, uint32_t k,
                   const float* __restrict__ vmax_sq, SearchCtl* __restrict__ ctl,
                   uint32_t* __restrict__ keys, uint32_t* __restrict__ tmax,
                   uint32_t* __restrict__ wmax, uint32_t* __restrict__ hist) {
    constexpr int RPW = 64 / G;   // rows per wave iteration
    constexpr int ITERS = G;      // iterations per 64-row tile
    constexpr int D4 = G * NV;    // float4 per row
    constexpr uint32_t D = 4u * D4;
    static_assert(ITERS % U == 0, "unroll must divide the tile");
    __shared__ v4f qs[D4];
    __shared__ uint32_t lhist[HIST_BINS];
    __shared__ float na_sh;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < (int)HIST_BINS; i += 256) lhist[i] = 0u;
    for (int i = tid; i < D4; i += 256) qs[i] = reinterpret_cast<const v4f*>(q)[i];
    // ||q|| with the reference's sequential sum, by one wave per workgroup:
    // needed for cosine scores, and published once for the final stage's
    // re-score and error bound
    if (wid == 0 && (METRIC == 0 || blockIdx.x == 0)) {
        const float nsq = query_norm_sq_wave(q, D, lane);
        if (lane == 0) {
            na_sh = sqrtf(nsq);
            if (blockIdx.x == 0) publish_ctl(ctl, nsq, need, k, vmax_sq, gridDim.x * 4u);
        }
    }
    __syncthreads();
    const int gl = lane % G, sub = lane / G;
    v4f qr[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) qr[j] = qs[j * G + gl];
    const float na = METRIC == 0 ? na_sh : 0.0f;
    const uint32_t n_tiles = (n_rows + 63u) >> 6;
    const uint32_t n_waves = gridDim.x * 4u;
    const uint32_t last = n_rows - 1u;

    uint32_t wave_best = 0u;  // largest key over all tiles of this wave
    for (uint32_t t = blockIdx.x * 4u + wid; t < n_tiles; t += n_waves) {
        const uint32_t row0 = t * 64u;
        float mine = 0.0f;
#pragma unroll 1
        for (int it0 = 0; it0 < ITERS; it0 += U) {
            v4f x[U][NV];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t r = min(row0 + (uint32_t)((it0 + u) * RPW + sub), last);
                const v4f* p = reinterpret_cast<const v4f*>(corpus + (size_t)r * D) + gl;
#pragma unroll
                for (int j = 0; j < NV; ++j) x[u][j] = __builtin_nontemporal_load(p + j * G);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
                for (int j = 0; j < NV; ++j) accum_fast<METRIC>(qr[j], x[u][j], a0, a1);
#pragma unroll
                for (int o = G / 2; o > 0; o >>= 1) {
                    a0 += __shfl_xor(a0, o);
                    if (METRIC == 0) a1 += __shfl_xor(a1, o);
                }
                const float s = finish_fast<METRIC>(a0, a1, na);
                if (it0 + u == gl) mine = s;
            }
        }
        // lane holds the score of tile row gl*RPW + sub
        const uint32_t row = row0 + (uint32_t)(gl * RPW + sub);
        uint32_t key = 0u;
        if (row < n_rows) {
            if (!row_is_dead(dead, row)) {
                key = score_key(mine);
                atomicAdd(&lhist[key >> 21], 1u);
            }
            keys[row] = key;
        }
        const uint32_t tm = wave_max_u32(key);
        if (lane == 0) tmax[t] = tm;
        wave_best = tm > wave_best ? tm : wave_best;
    }
    if (lane == 0) wmax[blockIdx.x * 4u + wid] = wave_best;
    __syncthreads();
    for (int i = tid; i < (int)HIST_BINS; i += 256) {
        const uint32_t h = lhist[i];
        if (h) atomicAdd(&hist[i], h);
    }
}

// Dimensions with a specialised kernel: d = 128*NV (G=32, NV in {1,3}) or
// d = 256*NV (G=64, NV in {1,2,3,4,6,8,12,16}).
bool stream_supports_dim(uint32_t d) {
    switch (d) {
    case 128: case 384:
    case 256: case 512: case 768: case 1024: case 1536: case 2048: case 3072: case 4096:
        return true;
    default:
        return false;
    }
}

template <int METRIC, int G, int NV, int U>
static uint32_t launch_one(const float* d_q, const ScanArgs& a, const Scratch& s, uint32_t need,
                       uint32_t k, const float* d_vmax_sq, hipStream_t st) {
    const uint32_t n_tiles = (a.n_rows + 63u) / 64u;
    uint32_t grid = (n_tiles + 3u) / 4u;
    // one resident wave set: workgroups per CU from the occupancy query, times the CU count
    static uint32_t cap = 0;
    if (cap == 0) {
        int per_cu = 0, dev = 0, cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(
                &per_cu, scan_stream_kernel<METRIC, G, NV, U>, 256, 0) != hipSuccess || per_cu < 1)
            per_cu = 4;
        if (hipGetDevice(&dev) == hipSuccess)
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        cap = (uint32_t)per_cu * (uint32_t)cus;
        if (cap * 4u > WMAX_CAP) cap = WMAX_CAP / 4u;
    }
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL((scan_stream_kernel<METRIC, G, NV, U>), dim3(grid), dim3(256), 0, st,
                       a.corpus, d_q, a.dead, a.n_rows, need, k, d_vmax_sq, s.ctl, s.keys, s.tmax, s.wmax,
                       s.hist);
    return grid * 4u;
}

template <int METRIC>
static uint32_t launch_metric(const float* d_q, const ScanArgs& a, const Scratch& s, uint32_t need,
                          uint32_t k, const float* d_vmax_sq, hipStream_t st) {
    switch (a.dim) {
    case 128: return launch_one<METRIC, 32, 1, 8>(d_q, a, s, need, k, d_vmax_sq, st);
    case 384: return launch_one<METRIC, 32, 3, 4>(d_q, a, s, need, k, d_vmax_sq, st);
    case 256: return launch_one<METRIC, 64, 1, 8>(d_q, a, s, need, k, d_vmax_sq, st);
    case 512: return launch_one<METRIC, 64, 2, 4>(d_q, a, s, need, k, d_vmax_sq, st);
    case 768: return launch_one<METRIC, 64, 3, 4>(d_q, a, s, need, k, d_vmax_sq, st);
    case 1024: return launch_one<METRIC, 64, 4, 4>(d_q, a, s, need, k, d_vmax_sq, st);
    case 1536: return launch_one<METRIC, 64, 6, 2>(d_q, a, s, need, k, d_vmax_sq, st);
    case 2048: return launch_one<METRIC, 64, 8, 2>(d_q, a, s, need, k, d_vmax_sq, st);
    case 3072: return launch_one<METRIC, 64, 12, 1>(d_q, a, s, need, k, d_vmax_sq, st);
    case 4096: return launch_one<METRIC, 64, 16, 1>(d_q, a, s, need, k, d_vmax_sq, st);
    default: return 0;
    }
}

hipError_t launch_scan_stream(const float* d_q, const ScanArgs& a, const Scratch& s, uint32_t need,
                              uint32_t k, const float* d_vmax_sq, uint32_t* n_waves, hipStream_t st) {
    if (!stream_supports_dim(a.dim)) return hipErrorInvalidValue;
    switch (a.metric) {
    case TRAG_COSINE: *n_waves = launch_metric<0>(d_q, a, s, need, k, d_vmax_sq, st); break;
    case TRAG_EUCLIDEAN: *n_waves = launch_metric<1>(d_q, a, s, need, k, d_vmax_sq, st); break;
    default: *n_waves = launch_metric<2>(d_q, a, s, need, k, d_vmax_sq, st); break;
    }
    return hipGetLastError();
}

}  // namespace trag
