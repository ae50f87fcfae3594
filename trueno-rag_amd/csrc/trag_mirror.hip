// Optional bf16 mirror of the corpus and the single-query scan that reads it.
//
// MI355X has 288 GB of HBM: a second copy of the corpus at half the width costs
// 50 % more memory and halves the bytes a prefiltering pass must read. The
// mirror never decides a result by itself: a pass over it yields approximate
// scores within a rigorous bound of the reference's f32 scores (mirror_eps in
// trag_device.h); the rows that can be in the top k under that bound are then
// scored from the f32 corpus with the reference arithmetic (select_mirror_kernel
// in trag_kernels.hip), so rows, ranks and score bits equal the exact scan's.
//
// scan_mirror_kernel: rows are read once with 16-byte non-temporal loads along
// the row (G lanes cover G*16 contiguous bytes = 8G bf16), the f32 query sits in
// registers, a bf16 becomes an f32 by a shift (exact), partial sums are combined
// with a wavefront butterfly. Per 64-row tile: one coalesced store of 64 ordered
// keys, the tile / wave maxima and 64 LDS histogram increments, as the other
// scans. Algorithmic bytes per launch: N * d * 2 (+ 4 N of squared norms for
// cosine, + 4 N of keys written).
//
// Where the time goes (MI355X, 10M x 384, 7.68 GB per launch; scripts/diag_stream.hip holds
// the standalone probes): the load pattern alone, with the conversions, FMAs and butterflies
// but no outputs, streams at 7.1 TB/s (1.08 ms). With every row's key stored this kernel takes
// 1.26 ms; compiling the outputs out one at a time: without the per-row key store 1.16 ms, also
// without the tile maxima 1.153, also without the norm / tombstone reads 1.149, also without
// the LDS histogram 1.147. The key store is 0.5 % of the bytes and 8 % of the time: the cost
// follows the number of tiles stored (every 4th tile: a quarter of it), not the instruction
// count or the batching (16 tiles buffered in LDS and flushed together: no change), and
// non-temporal or system-scope stores recover under half of it in the probe and nothing here.
// Hence the sparse key stores (trag_scan_common.h): 1.17 ms on long scans.
// Measured and rejected: an unconditional refill with a scheduling barrier after each load
// group (exactly 12 + 12 loads in flight, waits 23..12: 1.253 ms); a bound per coarse bin from
// each wave's own k best tile maxima shared through one word with atomic max (bins of a
// quarter binade plus the 2 eps margin leave ~3/4 of the tiles above the bound; polling the
// word every tile cost 17 %).
#include "trag_device.h"
#include "trag_internal.h"
#include "trag_scan_common.h"

namespace trag {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// f32 -> bf16, round to nearest even: the same conversion the matrix-core
// prefilter applies on the fly when it reads the f32 corpus.
__global__ void __launch_bounds__(256)
to_bf16_kernel(const float* __restrict__ src, unsigned long long n4, uint16_t* __restrict__ dst) {
    for (unsigned long long e = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; e < n4;
         e += (unsigned long long)gridDim.x * 256ull) {
        const v4f x = reinterpret_cast<const v4f*>(src)[e];
        reinterpret_cast<bf16x4*>(dst)[e] = __builtin_convertvector(x, bf16x4);
    }
}

hipError_t launch_to_bf16(const float* d_src, uint64_t n_floats, uint16_t* d_dst, hipStream_t st) {
    if (n_floats == 0) return hipSuccess;
    const uint64_t n4 = n_floats / 4;  // callers guarantee dim % 8 == 0
    uint64_t grid = (n4 + 256 * 8 - 1) / (256 * 8);
    if (grid > 256 * 16) grid = 256 * 16;
    hipLaunchKernelGGL(to_bf16_kernel, dim3((uint32_t)grid), dim3(256), 0, st, d_src, (unsigned long long)n4, d_dst);
    return hipGetLastError();
}

// eight bf16 of the row (one 16-byte piece) against eight f32 of the query
template <int METRIC>
__device__ __forceinline__ void accum_piece(const v4f qlo, const v4f qhi, const u32x4 w, float& a0, float& a1) {
    const float x0 = __uint_as_float(w.x << 16), x1 = __uint_as_float(w.x & 0xFFFF0000u);
    const float x2 = __uint_as_float(w.y << 16), x3 = __uint_as_float(w.y & 0xFFFF0000u);
    const float x4 = __uint_as_float(w.z << 16), x5 = __uint_as_float(w.z & 0xFFFF0000u);
    const float x6 = __uint_as_float(w.w << 16), x7 = __uint_as_float(w.w & 0xFFFF0000u);
    if (METRIC == 1) {
        float t;
        t = qlo.x - x0; a0 = __builtin_fmaf(t, t, a0);
        t = qlo.y - x1; a1 = __builtin_fmaf(t, t, a1);
        t = qlo.z - x2; a0 = __builtin_fmaf(t, t, a0);
        t = qlo.w - x3; a1 = __builtin_fmaf(t, t, a1);
        t = qhi.x - x4; a0 = __builtin_fmaf(t, t, a0);
        t = qhi.y - x5; a1 = __builtin_fmaf(t, t, a1);
        t = qhi.z - x6; a0 = __builtin_fmaf(t, t, a0);
        t = qhi.w - x7; a1 = __builtin_fmaf(t, t, a1);
    } else {
        a0 = __builtin_fmaf(qlo.x, x0, a0);
        a1 = __builtin_fmaf(qlo.y, x1, a1);
        a0 = __builtin_fmaf(qlo.z, x2, a0);
        a1 = __builtin_fmaf(qlo.w, x3, a1);
        a0 = __builtin_fmaf(qhi.x, x4, a0);
        a1 = __builtin_fmaf(qhi.y, x5, a1);
        a0 = __builtin_fmaf(qhi.z, x6, a0);
        a1 = __builtin_fmaf(qhi.w, x7, a1);
    }
}

// G lanes per row (16, 32 or 64), NV 16-byte pieces per lane and row, U row
// iterations in flight. d = 8 * G * NV.
template <int METRIC, int G, int NV, int U>
__global__ void __launch_bounds__(256)
scan_mirror_kernel(const uint16_t* __restrict__ mirror, const float* __restrict__ q,
                   const float* __restrict__ norm2, const uint32_t* __restrict__ dead, uint32_t n_rows,
                   uint32_t need, uint32_t k, const float* __restrict__ vmax_sq,
                   SearchCtl* __restrict__ ctl, uint32_t* __restrict__ keys, uint32_t* __restrict__ tmax,
                   uint32_t* __restrict__ wmax, uint32_t* __restrict__ hist,
                   unsigned long long* __restrict__ wbest, uint32_t epoch) {
    constexpr int ITERS = G;      // iterations per 64-row tile; iteration i takes rows i, G + i, 2G + i, ... of the tile
    constexpr int D16 = G * NV;   // 16-byte pieces per row
    constexpr uint32_t D = 8u * D16;
    static_assert(ITERS % U == 0, "unroll must divide the tile");
    __shared__ v4f qs[2 * D16];
    __shared__ uint32_t lhist[HIST_BINS];
    __shared__ float na_sh, nsq_sh;
    __shared__ uint32_t keep_lds;   // sparse key stores: the bound wave 0 of the workgroup last computed
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // sparse key stores (trag_scan_common.h): on when the launcher passes an epoch
    const bool track = epoch != 0u && k >= 1u && k <= 64u && gridDim.x * 4u <= SPARSE_WAVES;
    if (tid == 0) keep_lds = 0u;
    for (int i = tid; i < (int)HIST_BINS; i += 256) lhist[i] = 0u;
    for (int i = tid; i < 2 * D16; i += 256) qs[i] = reinterpret_cast<const v4f*>(q)[i];
    // ||q|| with the reference's sequential sum (needed for cosine scores here, and
    // published once for the final stage's exact scoring and error bound)
    if (wid == 0 && (METRIC == 0 || track || blockIdx.x == 0)) {
        const float nsq = query_norm_sq_wave(q, D, lane);
        if (lane == 0) {
            nsq_sh = nsq;
            na_sh = sqrtf(nsq);
            if (blockIdx.x == 0) publish_ctl(ctl, nsq, need, k, vmax_sq, gridDim.x * 4u);
        }
    }
    __syncthreads();
    const int gl = lane % G, sub = lane / G;
    v4f qlo[NV], qhi[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        qlo[j] = qs[2 * (j * G + gl)];
        qhi[j] = qs[2 * (j * G + gl) + 1];
    }
    const float na = METRIC == 0 ? na_sh : 0.0f;
    const uint32_t n_tiles = (n_rows + 63u) >> 6;
    const uint32_t n_waves = gridDim.x * 4u;
    const uint32_t last = n_rows - 1u;
    // The select lowers its threshold to (k-th best bounded approximate score) - 2.02 eps; the
    // keys kept reach a little further down, in the same arithmetic (0: no valid bound, keep all).
    float keep_margin = -1.0f;
    if (track) {
        const float nsq = nsq_sh;
        const float eps = mirror_eps<METRIC>(D, na_sh, sqrtf(*vmax_sq));
        if ((nsq == 0.0f || (nsq >= 0x1p-100f && nsq < 0x1p100f)) && eps < INFINITY) keep_margin = 2.02f * eps * 1.01f;
    }
    uint32_t keep_key = 0u;    // tiles whose best key is below this are not stored
    uint32_t keep_best = 0u;   // best bounded key this wave has seen
    uint32_t tiles_done = 0u;
    unsigned long long polled[16];

    // Two groups of U row-iterations ping-pong through registers: the loads of the next
    // group (of this tile or of the wave's next tile) are in flight while the current one
    // is reduced.
    static_assert((ITERS / U) % 2 == 0, "groups per tile must pair up");
    u32x4 xa[U][NV], xb[U][NV];
    auto issue = [&](u32x4 (&x)[U][NV], uint32_t tile, int it0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t r = min(tile * 64u + (uint32_t)(sub * ITERS + it0 + u), last);
            const u32x4* p = reinterpret_cast<const u32x4*>(mirror + (size_t)r * D) + gl;
#pragma unroll
            for (int j = 0; j < NV; ++j) x[u][j] = __builtin_nontemporal_load(p + j * G);
        }
    };
    float mine = 0.0f;
    auto reduce = [&](const u32x4 (&x)[U][NV], int it0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
            for (int j = 0; j < NV; ++j) accum_piece<METRIC>(qlo[j], qhi[j], x[u][j], a0, a1);
            a0 += a1;
#pragma unroll
            for (int o = G / 2; o > 0; o >>= 1) a0 += __shfl_xor(a0, o);
            if (it0 + u == gl) mine = a0;
        }
    };
    uint32_t wave_best = 0u;
    uint32_t t = blockIdx.x * 4u + wid;
    if (t < n_tiles) issue(xa, t, 0);
    for (; t < n_tiles; t += n_waves) {
        const uint32_t row0 = t * 64u;
        // this lane's row of the tile: its squared norm is fetched now and used after the loop
        const uint32_t row = row0 + (uint32_t)(sub * ITERS + gl);  // = row0 + lane
        float nb2 = 0.0f;
        if (METRIC == 0) nb2 = norm2[min(row, last)];
        const bool poll = track && wid == 0 && sparse_poll_at(tiles_done);   // one poller per workgroup
        if (poll) sparse_poll_issue(wbest, lane, polled);   // (used after the tile: the latency hides behind it)
#pragma unroll 1
        for (int it0 = 0; it0 < ITERS; it0 += 2 * U) {
            issue(xb, t, it0 + U);
            reduce(xa, it0);
            if (it0 + 2 * U < ITERS) issue(xa, t, it0 + 2 * U);
            else if (t + n_waves < n_tiles) issue(xa, t + n_waves, 0);
            reduce(xb, it0 + U);
        }
        // this lane holds the sum of tile row sub*G + gl = its own lane index
        uint32_t key = 0u;
        if (row < n_rows) {
            if (!row_is_dead(dead, row)) {
                // Rows outside the range in which the error bound holds (non-finite sums,
                // squared norm subnormal or infinite) get +inf: they always reach the
                // exact scoring, which decides their true rank.
                float s;
                bool ok = fabsf(mine) < INFINITY;
                if (METRIC == 0) {
                    ok = ok && (nb2 == 0.0f || (nb2 >= 0x1p-100f && nb2 < INFINITY));
                    const float nb = sqrtf(nb2);
                    s = (na == 0.0f || nb == 0.0f) ? 0.0f : mine / (na * nb);
                } else if (METRIC == 1) {
                    s = -sqrtf(mine);
                } else {
                    s = mine;
                }
                if (!ok) s = INFINITY;
                key = score_key(s);
                atomicAdd(&lhist[key >> 21], 1u);
            }
        }
        const uint32_t tm = wave_max_u32(key);
        if (track) {
            // (+inf marks rows without a valid bound: never one of the k bounded rows)
            if (tm >= 0x007FFFFFu && tm < 0xFF800000u && tm > keep_best) keep_best = tm;
            ++tiles_done;
            if (sparse_publish_at(tiles_done) && lane == 0)
                __hip_atomic_store(wbest + (blockIdx.x * 4u + wid), ((unsigned long long)epoch << 32) | keep_best,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (poll && keep_margin >= 0.0f) {
                const uint32_t kk = sparse_keep_key(sparse_poll_value(polled, epoch, k), keep_margin);
                if (kk > keep_key && lane == 0) __hip_atomic_store(&keep_lds, kk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                keep_key = kk > keep_key ? kk : keep_key;
            }
            // (the other waves pick the poller's bound up whenever it appears: any value it ever held is valid)
            const uint32_t shared_kk = __hip_atomic_load(&keep_lds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            keep_key = shared_kk > keep_key ? shared_kk : keep_key;
        }
        if (row < n_rows && tm >= keep_key) keys[row] = key;
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

// Dimensions with a specialised kernel: d = 8 * G * NV.
bool mirror_scan_supports_dim(uint32_t d) {
    switch (d) {
    case 128: case 256: case 384: case 512: case 768: case 1024: case 1536: case 2048: case 3072: case 4096:
        return true;
    default:
        return false;
    }
}

template <int METRIC, int G, int NV, int U>
static uint32_t launch_one(const uint16_t* mirror, const float* d_q, const float* d_norm2, const ScanArgs& a,
                           const Scratch& s, uint32_t need, uint32_t k, const float* d_vmax_sq, hipStream_t st) {
    const uint32_t n_tiles = (a.n_rows + 63u) / 64u;
    uint32_t grid = (n_tiles + 3u) / 4u;
    // one resident wave set: workgroups per CU from the occupancy query, times the CU count
    static uint32_t cap = 0;
    if (cap == 0) {
        int per_cu = 0, dev = 0, cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(
                &per_cu, scan_mirror_kernel<METRIC, G, NV, U>, 256, 0) != hipSuccess || per_cu < 1)
            per_cu = 2;
        // measured on MI355X (10M x 384): 1 workgroup per CU 6.28 TB/s, 2: 6.07, 4 and more: 6.0
        if (per_cu > 1) per_cu = 1;
        if (hipGetDevice(&dev) == hipSuccess)
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        cap = (uint32_t)per_cu * (uint32_t)cus;
        if (cap * 4u > WMAX_CAP) cap = WMAX_CAP / 4u;
    }
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL((scan_mirror_kernel<METRIC, G, NV, U>), dim3(grid), dim3(256), 0, st, mirror, d_q,
                       d_norm2, a.dead, a.n_rows, need, k, d_vmax_sq, s.ctl, s.keys, s.tmax, s.wmax, s.hist,
                       s.wbest, sparse_scan_pays(n_tiles, grid * 4u) && s.wbest ? s.epoch : 0u);
    return grid * 4u;
}

template <int METRIC>
static uint32_t launch_metric(const uint16_t* mirror, const float* d_q, const float* d_norm2, const ScanArgs& a,
                              const Scratch& s, uint32_t need, uint32_t k, const float* d_vmax_sq, hipStream_t st) {
#define TRAG_MIRROR_CASE(DIM, G, NV, U) \
    case DIM: return launch_one<METRIC, G, NV, U>(mirror, d_q, d_norm2, a, s, need, k, d_vmax_sq, st)
    switch (a.dim) {
        TRAG_MIRROR_CASE(128, 16, 1, 8);
        TRAG_MIRROR_CASE(256, 32, 1, 8);
        TRAG_MIRROR_CASE(384, 16, 3, 4);  // (8 lanes per row, or 2 / 8 rows in flight: within 2 %)
        TRAG_MIRROR_CASE(512, 64, 1, 8);
        TRAG_MIRROR_CASE(768, 32, 3, 4);
        TRAG_MIRROR_CASE(1024, 64, 2, 4);
        TRAG_MIRROR_CASE(1536, 64, 3, 4);
        TRAG_MIRROR_CASE(2048, 64, 4, 2);
        TRAG_MIRROR_CASE(3072, 64, 6, 2);
        TRAG_MIRROR_CASE(4096, 64, 8, 2);
    default: return 0;
    }
#undef TRAG_MIRROR_CASE
}

hipError_t launch_scan_mirror(const uint16_t* d_mirror, const float* d_q, const float* d_norm2, const ScanArgs& a,
                              const Scratch& s, uint32_t need, uint32_t k, const float* d_vmax_sq,
                              uint32_t* n_waves, hipStream_t st) {
    if (!mirror_scan_supports_dim(a.dim)) return hipErrorInvalidValue;
    switch (a.metric) {
    case TRAG_COSINE: *n_waves = launch_metric<0>(d_mirror, d_q, d_norm2, a, s, need, k, d_vmax_sq, st); break;
    case TRAG_EUCLIDEAN: *n_waves = launch_metric<1>(d_mirror, d_q, d_norm2, a, s, need, k, d_vmax_sq, st); break;
    default: *n_waves = launch_metric<2>(d_mirror, d_q, d_norm2, a, s, need, k, d_vmax_sq, st); break;
    }
    return hipGetLastError();
}

}  // namespace trag
