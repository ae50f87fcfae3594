// gfx950 kernels of the dense-retrieval path: query preparation, the exact
// corpus scan, the radix-select stages and the final sort.
//
// This file is compiled with -ffp-contract=off: every `a * b` followed by `+`
// below is two roundings, exactly as the reference's scalar f32 loops
// (src/index.rs:440-462 of the reference). Fused multiply-adds appear only
// where written as __builtin_fmaf.
#include <algorithm>

#include "trag_device.h"
#include "trag_internal.h"
#include "trag_scan_common.h"

namespace trag {

// ---------------------------------------------------------------------------
// Score arithmetic, reference order. METRIC: 0 cosine, 1 euclidean, 2 dot.
// acc0 is the dot (or squared distance) accumulator, acc1 the row's sum of squares.
// ---------------------------------------------------------------------------
template <int METRIC>
__device__ __forceinline__ void accum(float qv, float x, float& acc0, float& acc1) {
    if (METRIC == 1) {
        float t = qv - x;      // (x - y)
        float p = t * t;       // .powi(2)
        acc0 = acc0 + p;
    } else {
        float p = qv * x;
        acc0 = acc0 + p;
        if (METRIC == 0) {
            float p2 = x * x;
            acc1 = acc1 + p2;
        }
    }
}

template <int METRIC>
__device__ __forceinline__ float finish_score(float acc0, float acc1, float na) {
    if (METRIC == 0) {
        float nb = sqrtf(acc1);
        if (na == 0.0f || nb == 0.0f) return 0.0f;   // src/index.rs:445-446
        float den = na * nb;
        return acc0 / den;                           // src/index.rs:448
    } else if (METRIC == 1) {
        return -sqrtf(acc0);                         // src/index.rs:400,452-458
    } else {
        return acc0;                                 // src/index.rs:460-462
    }
}

// ---------------------------------------------------------------------------
// Exact scan, d % 4 == 0.
//
// One lane owns one row, so its sums run left to right exactly as the
// reference's. To keep HBM reads coalesced each wave stages a tile of 64 rows x
// 32 columns through a private, padded LDS image: 16-byte loads walk along the
// rows (8 lanes cover 128 contiguous bytes of one row), the image is written
// row-major with a 36-float stride, and every lane then reads its own row with
// conflict-free ds_read_b128. Tiles are private to a wave: no workgroup
// barrier in the loop. The next chunk's global loads are in flight while the
// current chunk is reduced. Each row's ordered key goes to keys[], and a
// workgroup-local histogram of the keys' top 11 bits is merged into hist[].
//
// Ceiling: the same load pattern with no LDS and no arithmetic reads 7.10 TB/s on
// MI355X (a plain contiguous stream 7.33); this kernel reaches 6.65 (10M x 384).
// Measured and rejected: two chunks of loads in flight per wave (2 % slower);
// running the one-workgroup select inside this kernel's
// last-arriving workgroup (agent-scope release + ticket per workgroup). The
// 256 release fences and a 256-thread select cost more than the launch boundary
// they remove: step 0.278 ms vs 0.256 ms at 1M x 384, 2.357 vs 2.329 at 10M x 384.
// With the histogram atomics, the keys[] stores and the tombstone reads all compiled
// out the scan takes 2.264 ms instead of 2.330 (10M x 384); the keys[] stores are most
// of that, and long scans skip the ones no later stage can read (sparse key stores,
// trag_scan_common.h: 2.262 ms against 2.382 on the same, slower, device).
// With those in place the per-tile maximum store costs nothing measurable (compiled out:
// 2.270 / 2.264 ms with, 2.297 / 2.272 without), and 64-float chunks do not help long rows
// either (2M x 4096: 5.054 ms with 32-float chunks, 5.062 with 64; 1M x 1536: 0.926 / 0.935).
// ---------------------------------------------------------------------------
// (defined with the select kernels below)
template <int METRIC, bool RESCORE, int NT>
__device__ __forceinline__ void
select_final_body(const float* __restrict__ corpus, const float* __restrict__ q, uint32_t n_rows,
                  uint32_t d, uint32_t live, uint32_t need, uint32_t k, uint32_t n_waves,
                  SearchCtl* __restrict__ ctl, uint32_t* __restrict__ hist,
                  const uint32_t* __restrict__ keys, const uint32_t* __restrict__ tmax,
                  const uint32_t* __restrict__ wmax, uint32_t* __restrict__ out_rows,
                  float* __restrict__ out_scores, uint32_t* __restrict__ out_n,
                  uint32_t* __restrict__ out_status, uint32_t* __restrict__ events,
                  const unsigned long long* __restrict__ wbest, uint32_t epoch,
                  unsigned long long* sk, uint32_t* hit, uint32_t* sh, uint32_t* cnt);

template <int METRIC, int EX_CW /* chunk width in floats: 32 or 64 */, bool NT, int WPB /* waves per workgroup */,
          bool ROLL /* refill load by load instead of chunk by chunk */,
          bool FUSE = false /* workgroup 0 runs the one-workgroup select of the handle's PREVIOUS query */>
__global__ void __launch_bounds__(WPB * 64)
scan_exact_v4_kernel(const float* __restrict__ corpus, const float* __restrict__ q,
                     const uint32_t* __restrict__ dead, uint32_t n_rows, uint32_t d,
                     uint32_t need, uint32_t k, const float* __restrict__ vmax_sq,
                     SearchCtl* __restrict__ ctl, uint32_t* __restrict__ keys,
                     uint32_t* __restrict__ tmax, uint32_t* __restrict__ wmax,
                     uint32_t* __restrict__ hist, unsigned long long* __restrict__ wbest, uint32_t epoch,
                     PrevSelect ps) {
    constexpr int EX_STRIDE = EX_CW + 4;  // padded LDS row stride in floats
    constexpr int LPR = EX_CW / 4;        // lanes that cover one row's chunk with 16-byte loads
    constexpr int RPL = 64 / LPR;         // rows covered by one wave-wide load
    constexpr int NLD = 64 / RPL;         // loads per lane per chunk
    __shared__ __attribute__((aligned(16))) float tiles[WPB][64 * EX_STRIDE];
    __shared__ uint32_t lhist[HIST_BINS];
    __shared__ float na_sh;
    __shared__ uint32_t keep_lds;   // sparse key stores: the bound wave 0 of the workgroup last computed
    // FUSE: one launch = this query's scan + the previous query's select. The select is one workgroup's
    // work of some 20 us, during which the rest of the device would idle if it ran as a kernel of its
    // own behind the scan; here it is workgroup 0 of the NEXT scan's launch (the scan's workgroups are
    // 1 .. gridDim.x - 1), reads the other scratch set, and costs the stream nothing: no second launch,
    // no event, no second stream. Its candidate buffer lies over the tile images (a workgroup is one
    // or the other).
    uint32_t bid = blockIdx.x, nblk = gridDim.x;
    if (FUSE) {
        __shared__ uint32_t sel_hit[FUSE ? CAND_CAP : 1];
        __shared__ uint32_t sel_sh[FUSE ? 24 : 1], sel_cnt[FUSE ? 4 : 1];
        static_assert(!FUSE || sizeof(tiles) >= CAND_CAP * sizeof(unsigned long long), "candidate keys fit the tile images");
        if (blockIdx.x == 0) {
            select_final_body<METRIC, false, WPB * 64>(nullptr, ps.q, ps.n_rows, ps.d, ps.live, ps.need, ps.k, ps.n_waves,
                                                       ps.ctl, ps.hist, ps.keys, ps.tmax, ps.wmax, ps.out_rows,
                                                       ps.out_scores, ps.out_n, ps.out_status, ps.events, ps.wbest,
                                                       ps.epoch, reinterpret_cast<unsigned long long*>(&tiles[0][0]),
                                                       sel_hit, sel_sh, sel_cnt);
            return;
        }
        bid -= 1u;
        nblk -= 1u;
    }
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) keep_lds = 0u;
    for (int i = tid; i < (int)HIST_BINS; i += WPB * 64) lhist[i] = 0u;

    float* tile = tiles[wid];
    const uint32_t n_tiles = (n_rows + 63u) >> 6;
    const uint32_t n_waves = nblk * (uint32_t)WPB;
    const uint32_t d4 = d >> 2;
    const uint32_t n_chunks = (d + EX_CW - 1) / EX_CW;
    const int lrow = lane / LPR;  // row of a load within its group of RPL
    const int lcol = lane % LPR;  // float4 column within the chunk

    uint32_t t = bid * (uint32_t)WPB + wid;
    v4f pre[NLD];
    {
        // issue the loads of chunk c of tile tt
        auto issue = [&](uint32_t tt, uint32_t c) {
            const uint32_t col4 = min(c * (uint32_t)LPR + (uint32_t)lcol, d4 - 1u);
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const uint32_t r = min(tt * 64u + (uint32_t)(i * RPL + lrow), n_rows - 1u);
                const v4f* p = reinterpret_cast<const v4f*>(corpus + (size_t)r * d) + col4;
                pre[i] = NT ? __builtin_nontemporal_load(p) : *p;
            }
        };
        if (t < n_tiles) issue(t, 0);
        // ||q|| while the first loads are in flight: one wave per workgroup runs
        // the sequential sum (cosine needs it in every workgroup; for the other
        // metrics only the copy published for the final stage is used)
        if (wid == 0 && (METRIC == 0 || bid == 0)) {
            const float nsq = query_norm_sq_wave(q, d, lane);
            if (lane == 0) {
                na_sh = sqrtf(nsq);
                if (bid == 0) publish_ctl(ctl, nsq, need, k, vmax_sq, n_waves);
            }
        }
        __syncthreads();  // also covers the histogram zeroing above
        const float na = METRIC == 0 ? na_sh : 0.0f;
        uint32_t wave_best = 0u;  // largest key over all tiles of this wave
        const uint32_t wave_id = bid * (uint32_t)WPB + wid;
        // sparse key stores (trag_scan_common.h): exact keys, so the bound is the need-th
        // largest published key itself
        const bool track = epoch != 0u && need >= 1u && need <= 64u && n_waves <= SPARSE_WAVES;
        uint32_t keep_key = 0u, tiles_done = 0u;
        unsigned long long polled[16];
        while (t < n_tiles) {
            float acc0 = 0.0f, acc1 = 0.0f;
            const bool poll = track && wid == 0 && sparse_poll_at(tiles_done);   // one poller per workgroup
            if (poll) sparse_poll_issue(wbest, lane, polled);   // (used after the tile)
            for (uint32_t c = 0; c < n_chunks; ++c) {
                // Stage chunk c (the reads of the previous chunk are older DS ops of this wave) and
                // refill. ROLL: as soon as load i of this chunk has landed and gone to LDS, load i of
                // the next chunk (of this tile or of the wave's next tile; past the end: a clamped,
                // harmless address) is issued into the same registers, so the wave keeps NLD loads in
                // flight instead of draining to zero before it issues again (used for short rows only,
                // see the launcher).
                const bool same = c + 1 < n_chunks;
                const uint32_t nt = same ? t : t + n_waves, nc = same ? c + 1 : 0u;
                const uint32_t ncol4 = min(nc * (uint32_t)LPR + (uint32_t)lcol, d4 - 1u);
#pragma unroll
                for (int i = 0; i < NLD; ++i) {
                    *reinterpret_cast<v4f*>(&tile[(i * RPL + lrow) * EX_STRIDE + lcol * 4]) = pre[i];
                    if (ROLL) {
                        const uint32_t r = min(nt * 64u + (uint32_t)(i * RPL + lrow), n_rows - 1u);
                        const v4f* p = reinterpret_cast<const v4f*>(corpus + (size_t)r * d) + ncol4;
                        pre[i] = NT ? __builtin_nontemporal_load(p) : *p;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (!ROLL) {  // chunk by chunk: all loads of the next chunk after the staging
                    if (same) issue(t, c + 1);
                    else if (t + n_waves < n_tiles) issue(t + n_waves, 0);
                }
                const float* __restrict__ qc = q + c * EX_CW;
                const float* myrow = tile + lane * EX_STRIDE;
                const uint32_t len = min((uint32_t)EX_CW, d - c * EX_CW);
                if (len == EX_CW) {
#pragma unroll
                    for (int j = 0; j < EX_CW / 4; ++j) {
                        const v4f x = *reinterpret_cast<const v4f*>(myrow + j * 4);
                        accum<METRIC>(qc[j * 4 + 0], x.x, acc0, acc1);
                        accum<METRIC>(qc[j * 4 + 1], x.y, acc0, acc1);
                        accum<METRIC>(qc[j * 4 + 2], x.z, acc0, acc1);
                        accum<METRIC>(qc[j * 4 + 3], x.w, acc0, acc1);
                    }
                } else {
                    for (uint32_t j = 0; j < len / 4; ++j) {
                        const v4f x = *reinterpret_cast<const v4f*>(myrow + j * 4);
                        accum<METRIC>(qc[j * 4 + 0], x.x, acc0, acc1);
                        accum<METRIC>(qc[j * 4 + 1], x.y, acc0, acc1);
                        accum<METRIC>(qc[j * 4 + 2], x.z, acc0, acc1);
                        accum<METRIC>(qc[j * 4 + 3], x.w, acc0, acc1);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            const uint32_t row = t * 64u + lane;
            uint32_t key = 0u;
            if (row < n_rows) {
                if (!row_is_dead(dead, row)) {
                    key = score_key(finish_score<METRIC>(acc0, acc1, na));
                    atomicAdd(&lhist[key >> 21], 1u);
                }
            }
            const uint32_t tm = wave_max_u32(key);
            wave_best = tm > wave_best ? tm : wave_best;
            if (track) {
                ++tiles_done;
                if (sparse_publish_at(tiles_done) && lane == 0)
                    __hip_atomic_store(wbest + wave_id, ((unsigned long long)epoch << 32) | wave_best,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (poll) {
                    const uint32_t kk = sparse_poll_value(polled, epoch, need);
                    if (kk > keep_key && lane == 0) __hip_atomic_store(&keep_lds, kk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    keep_key = kk > keep_key ? kk : keep_key;
                }
                // (the other waves pick the poller's bound up whenever it appears: any value it ever held is valid)
                const uint32_t shared_kk = __hip_atomic_load(&keep_lds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                keep_key = shared_kk > keep_key ? shared_kk : keep_key;
            }
            if (row < n_rows && tm >= keep_key) keys[row] = key;
            if (lane == 0) {
                tmax[t] = tm;
            }
            t += n_waves;
        }
        if (lane == 0) wmax[wave_id] = wave_best;
    }
    __syncthreads();
    for (int i = tid; i < (int)HIST_BINS; i += WPB * 64) {
        const uint32_t h = lhist[i];
        if (h) atomicAdd(&hist[i], h);
    }
}

// ---------------------------------------------------------------------------
// Exact scan of NQ queries in one corpus pass (d % 4 == 0). Same staging as
// above; every lane keeps one accumulator per query for its row, so each
// (row, query) sum still runs left to right in the reference's order, and the
// row's own sum of squares (cosine) is computed once for all queries. The
// pass is still bound by the HBM read of the corpus: NQ queries for the bytes
// of one. Query j's keys / maxima / histogram / control block live at
// j * stride behind the base pointers (MultiStride).
// ---------------------------------------------------------------------------
template <int METRIC, int NQ>
__global__ void __launch_bounds__(256)
scan_exact_multi_kernel(const float* __restrict__ corpus, const float* __restrict__ q,
                        const uint32_t* __restrict__ dead, uint32_t n_rows, uint32_t d,
                        uint32_t need, uint32_t k, const float* __restrict__ vmax_sq,
                        SearchCtl* __restrict__ ctl, uint32_t* __restrict__ keys,
                        uint32_t* __restrict__ tmax, uint32_t* __restrict__ wmax,
                        uint32_t* __restrict__ hist, MultiStride ms) {
    constexpr int EX_CW = 32, EX_STRIDE = EX_CW + 4, LPR = EX_CW / 4, RPL = 64 / LPR, NLD = 64 / RPL, WPB = 4;
    __shared__ __attribute__((aligned(16))) float tiles[WPB][64 * EX_STRIDE];
    __shared__ uint32_t lhist[NQ][HIST_BINS];
    __shared__ float na_sh[NQ];
    // the NQ queries, contiguous (NQ * d floats, dynamic): every lane reads them with
    // same-address (broadcast) 16-byte LDS loads, which keeps the vector-memory queue
    // for the corpus prefetch alone and the scalar registers out of the picture
    extern __shared__ __attribute__((aligned(16))) float qsh[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < NQ * (int)HIST_BINS; i += WPB * 64) (&lhist[0][0])[i] = 0u;

    float* tile = tiles[wid];
    const uint32_t n_tiles = (n_rows + 63u) >> 6;
    const uint32_t n_waves = gridDim.x * (uint32_t)WPB;
    const uint32_t d4 = d >> 2;
    const uint32_t n_chunks = (d + EX_CW - 1) / EX_CW;
    const int lrow = lane / LPR, lcol = lane % LPR;

    uint32_t t = blockIdx.x * (uint32_t)WPB + wid;
    v4f pre[NLD];
    auto issue = [&](uint32_t tt, uint32_t c) {
        const uint32_t col4 = min(c * (uint32_t)LPR + (uint32_t)lcol, d4 - 1u);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const uint32_t r = min(tt * 64u + (uint32_t)(i * RPL + lrow), n_rows - 1u);
            pre[i] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(corpus + (size_t)r * d) + col4);
        }
    };
    if (t < n_tiles) issue(t, 0);
    for (uint32_t i = tid; i < (uint32_t)NQ * d4; i += WPB * 64)
        reinterpret_cast<v4f*>(qsh)[i] = reinterpret_cast<const v4f*>(q)[i];
    // ||q_j|| while the first loads are in flight: wave w takes queries w, w + 4, ...
    for (int j = wid; j < NQ; j += WPB) {
        if (METRIC == 0 || blockIdx.x == 0) {
            const float nsq = query_norm_sq_wave(q + (size_t)j * d, d, lane);
            if (lane == 0) {
                na_sh[j] = sqrtf(nsq);
                if (blockIdx.x == 0)
                    publish_ctl(reinterpret_cast<SearchCtl*>(reinterpret_cast<char*>(ctl) + j * ms.zero_bytes),
                                nsq, need, k, vmax_sq, n_waves);
            }
        }
    }
    __syncthreads();
    float na[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) na[j] = METRIC == 0 ? na_sh[j] : 0.0f;
    uint32_t wave_best[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) wave_best[j] = 0u;
    const uint32_t wave_id = blockIdx.x * (uint32_t)WPB + wid;
    while (t < n_tiles) {
        float acc0[NQ], acc1 = 0.0f;
#pragma unroll
        for (int j = 0; j < NQ; ++j) acc0[j] = 0.0f;
        for (uint32_t c = 0; c < n_chunks; ++c) {
#pragma unroll
            for (int i = 0; i < NLD; ++i)
                *reinterpret_cast<v4f*>(&tile[(i * RPL + lrow) * EX_STRIDE + lcol * 4]) = pre[i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (c + 1 < n_chunks) issue(t, c + 1);
            else if (t + n_waves < n_tiles) issue(t + n_waves, 0);
            const float* qc = qsh + c * EX_CW;
            const float* myrow = tile + lane * EX_STRIDE;
            const uint32_t len = min((uint32_t)EX_CW, d - c * EX_CW);
            // four consecutive columns of the row against the same columns of every query
            auto quad = [&](uint32_t col) {
                const v4f x = *reinterpret_cast<const v4f*>(myrow + col);
                v4f qq[NQ];
#pragma unroll
                for (int j = 0; j < NQ; ++j) qq[j] = *reinterpret_cast<const v4f*>(qc + (size_t)j * d + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xe = x[e];
                    if (METRIC == 0) {
                        const float p2 = xe * xe;
                        acc1 = acc1 + p2;
                    }
#pragma unroll
                    for (int j = 0; j < NQ; ++j) {
                        const float qv = qq[j][e];
                        if (METRIC == 1) {
                            const float tdiff = qv - xe;
                            const float p = tdiff * tdiff;
                            acc0[j] = acc0[j] + p;
                        } else {
                            const float p = qv * xe;
                            acc0[j] = acc0[j] + p;
                        }
                    }
                }
            };
            if (len == EX_CW) {
#pragma unroll
                for (int j4 = 0; j4 < EX_CW / 4; ++j4) quad(j4 * 4);
            } else {
                for (uint32_t j4 = 0; j4 < len / 4; ++j4) quad(j4 * 4);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        const uint32_t row = t * 64u + lane;
        const bool live = row < n_rows && !row_is_dead(dead, row);
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            uint32_t key = 0u;
            if (live) {
                key = score_key(finish_score<METRIC>(acc0[j], acc1, na[j]));
                atomicAdd(&lhist[j][key >> 21], 1u);
            }
            if (row < n_rows) keys[(size_t)j * ms.keys + row] = key;
            const uint32_t tm = wave_max_u32(key);
            if (lane == 0) tmax[(size_t)j * ms.tmax + t] = tm;
            wave_best[j] = tm > wave_best[j] ? tm : wave_best[j];
        }
        t += n_waves;
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) wmax[(size_t)j * ms.wmax + wave_id] = wave_best[j];
    }
    __syncthreads();
    for (int i = tid; i < NQ * (int)HIST_BINS; i += WPB * 64) {
        const uint32_t h = (&lhist[0][0])[i];
        if (h) {
            const int j = i / (int)HIST_BINS, bin = i % (int)HIST_BINS;
            atomicAdd(reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(hist) + j * ms.zero_bytes) + bin, h);
        }
    }
}

// Exact scan for any d (rows need no alignment): one thread per row, strided
// loads. Used for dimensions that are not a multiple of 4 (toy sizes).
template <int METRIC>
__global__ void __launch_bounds__(256)
scan_exact_any_kernel(const float* __restrict__ corpus, const float* __restrict__ q,
                      const uint32_t* __restrict__ dead, uint32_t n_rows, uint32_t d,
                      uint32_t need, uint32_t k, const float* __restrict__ vmax_sq,
                      SearchCtl* __restrict__ ctl, uint32_t* __restrict__ keys,
                      uint32_t* __restrict__ tmax, uint32_t* __restrict__ wmax,
                      uint32_t* __restrict__ hist) {
    __shared__ uint32_t lhist[HIST_BINS];
    const int tid = threadIdx.x;
    for (int i = tid; i < (int)HIST_BINS; i += 256) lhist[i] = 0u;
    __syncthreads();
    const float nsq = query_norm_sq_wave(q, d, tid & 63);
    const float na = sqrtf(nsq);
    if (blockIdx.x == 0 && tid == 0) publish_ctl(ctl, nsq, need, k, vmax_sq, gridDim.x * 4u);
    uint32_t wave_best = 0u;
    // each wave covers one aligned 64-row tile per iteration
    for (uint32_t base = blockIdx.x * 256u; base < n_rows; base += gridDim.x * 256u) {
        const uint32_t row = base + tid;
        uint32_t key = 0u;
        if (row < n_rows && !row_is_dead(dead, row)) {
            const float* v = corpus + (size_t)row * d;
            float acc0 = 0.0f, acc1 = 0.0f;
            for (uint32_t j = 0; j < d; ++j) accum<METRIC>(q[j], v[j], acc0, acc1);
            key = score_key(finish_score<METRIC>(acc0, acc1, na));
            atomicAdd(&lhist[key >> 21], 1u);
        }
        if (row < n_rows) keys[row] = key;
        const uint32_t tm = wave_max_u32(key);
        const uint32_t tile = (base + (tid & ~63u)) >> 6;
        if ((tid & 63) == 0 && (base + (tid & ~63u)) < n_rows) tmax[tile] = tm;
        wave_best = tm > wave_best ? tm : wave_best;
    }
    if ((tid & 63) == 0) wmax[blockIdx.x * 4u + (tid >> 6)] = wave_best;
    __syncthreads();
    for (int i = tid; i < (int)HIST_BINS; i += 256) {
        const uint32_t h = lhist[i];
        if (h) atomicAdd(&hist[i], h);
    }
}

// ---------------------------------------------------------------------------
// Radix select over keys[]: find the threshold key such that at least `need`
// and at most CAND_CAP rows lie at or above it.
// ---------------------------------------------------------------------------

// Called by all NT threads of a workgroup (NT = 256 or 1024). hist has
// HIST_BINS bins. Returns the highest bin b whose suffix count (bins >= b)
// reaches `need`, and the count strictly above it. sh: NT/64 + 2 words of LDS.
template <int NT>
__device__ void find_bin(const uint32_t* __restrict__ hist, uint32_t need, uint32_t& bin,
                         uint32_t& above, uint32_t* sh, uint32_t* in_bin = nullptr) {
    constexpr int BPT = (int)HIST_BINS / NT;  // bins per thread
    constexpr int NW = NT / 64;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int top = (int)HIST_BINS - 1 - BPT * t;  // thread t owns bins top, top-1, ...
    uint32_t h[BPT], sum = 0;
#pragma unroll
    for (int i = 0; i < BPT; ++i) { h[i] = hist[top - i]; sum += h[i]; }
    uint32_t incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) sh[w] = incl;
    if (t == 0) { sh[NW] = 0; sh[NW + 1] = 0; sh[NW + 2] = hist[0]; }
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (int i = 0; i < NW; ++i) {
        if (i < w) base += sh[i];
        total += sh[i];
    }
    incl += base;
    const uint32_t excl = incl - sum;
    if (need > total) need = total;
    if (need > 0 && excl < need && need <= incl) {
        uint32_t cum = excl;
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            if (cum + h[i] >= need) { sh[NW] = (uint32_t)(top - i); sh[NW + 1] = cum; sh[NW + 2] = h[i]; break; }
            cum += h[i];
        }
    }
    __syncthreads();
    bin = sh[NW];
    above = sh[NW + 1];
    if (in_bin) *in_bin = sh[NW + 2];  // rows in that bin (saves the caller a dependent global load)
    __syncthreads();
}

// Level 2 (bits 20..10) and level 3 (bits 9..0) histogram passes. Each starts
// by resolving the previous level; when the rows at or above that level's bin
// already fit CAND_CAP the threshold is fixed and the pass is skipped.
template <int LEVEL>
__global__ void __launch_bounds__(256)
refine_kernel(const uint32_t* __restrict__ keys, uint32_t n_rows, SearchCtl* __restrict__ ctl,
              uint32_t* __restrict__ hist) {
    __shared__ uint32_t lhist[HIST_BINS];
    __shared__ uint32_t sh[8];
    const int tid = threadIdx.x;
    // Workgroup 0 of this very launch may set `resolved` while others start:
    // take one reading per workgroup so that all its threads agree.
    if (tid == 0) sh[6] = ctl->resolved;
    __syncthreads();
    if (sh[6]) return;
    const uint32_t need = ctl->need;
    uint32_t prefix, shift_match, shift_bin, mask_bin;
    if (LEVEL == 2) {
        uint32_t b1, above1;
        find_bin<256>(hist, need, b1, above1, sh);
        const uint32_t inbin = hist[b1];
        if (above1 + inbin <= CAND_CAP) {
            if (blockIdx.x == 0 && tid == 0) {
                ctl->t_key = b1 << 21;
                ctl->expect = above1 + inbin;
                ctl->resolved = 1;
            }
            return;
        }
        if (blockIdx.x == 0 && tid == 0) { ctl->b1 = b1; ctl->above1 = above1; }
        prefix = b1; shift_match = 21; shift_bin = 10; mask_bin = 2047u;
    } else {
        const uint32_t b1 = ctl->b1, above1 = ctl->above1;
        uint32_t b2, a2;
        find_bin<256>(hist + HIST_BINS, need - above1, b2, a2, sh);
        const uint32_t inbin = hist[HIST_BINS + b2];
        const uint32_t above2 = above1 + a2;
        if (above2 + inbin <= CAND_CAP) {
            if (blockIdx.x == 0 && tid == 0) {
                ctl->t_key = (b1 << 21) | (b2 << 10);
                ctl->expect = above2 + inbin;
                ctl->resolved = 2;
            }
            return;
        }
        if (blockIdx.x == 0 && tid == 0) { ctl->b2 = b2; ctl->above2 = above2; }
        prefix = (b1 << 11) | b2; shift_match = 10; shift_bin = 0; mask_bin = 1023u;
    }
    for (int i = tid; i < (int)HIST_BINS; i += 256) lhist[i] = 0u;
    __syncthreads();
    for (uint32_t i = blockIdx.x * 256u + tid; i < n_rows; i += gridDim.x * 256u) {
        const uint32_t key = keys[i];
        if (key != 0u && (key >> shift_match) == prefix)
            atomicAdd(&lhist[(key >> shift_bin) & mask_bin], 1u);
    }
    __syncthreads();
    uint32_t* out = hist + (LEVEL - 1) * HIST_BINS;
    for (int i = tid; i < (int)HIST_BINS; i += 256) {
        const uint32_t h = lhist[i];
        if (h) atomicAdd(&out[i], h);
    }
}

// Collect the candidate rows. Rows strictly above the threshold (or at/above
// it when they all fit) are appended in arbitrary order; the final stage sorts
// them, so the result does not depend on that order. When more than CAND_CAP
// rows tie on the threshold key itself, workgroup 0 adds exactly tie_quota of
// them in ascending row order.
__global__ void __launch_bounds__(256)
compact_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ tmax, uint32_t n_rows,
               SearchCtl* __restrict__ ctl, const uint32_t* __restrict__ hist,
               uint32_t* __restrict__ cand, size_t keys_stride, size_t tmax_stride) {
    // blockIdx.y = query of a batch (strides are 0 for a single query)
    keys += (size_t)blockIdx.y * keys_stride;
    tmax += (size_t)blockIdx.y * tmax_stride;
    ctl += blockIdx.y;
    cand += (size_t)blockIdx.y * CAND_CAP;
    __shared__ uint32_t sh[8];
    __shared__ uint32_t wcnt[4];
    __shared__ uint32_t taken_sh;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    uint32_t t_key, cut = 0, quota = 0, above3 = 0;
    if (ctl->resolved) {
        t_key = ctl->t_key;
    } else {
        const uint32_t need = ctl->need;
        const uint32_t b1 = ctl->b1, b2 = ctl->b2, above2 = ctl->above2;
        uint32_t b3, a3;
        find_bin<256>(hist + 2 * HIST_BINS, need - above2, b3, a3, sh);
        const uint32_t inbin = hist[2 * HIST_BINS + b3];
        above3 = above2 + a3;
        t_key = (b1 << 21) | (b2 << 10) | b3;
        uint32_t expect = above3 + inbin;
        if (expect > CAND_CAP) { cut = 1; quota = need - above3; expect = need; }
        if (blockIdx.x == 0 && tid == 0) {
            ctl->t_key = t_key;
            ctl->cut_ties = cut;
            ctl->tie_quota = quota;
            ctl->above3 = above3;
            ctl->expect = expect;
        }
    }
    const uint32_t t_min = t_key > 1u ? t_key : 1u;  // key 0 marks removed rows
    // Only tiles whose largest key reaches the threshold can hold a candidate:
    // each lane tests one tile maximum, then the wave visits the tiles that pass.
    const uint32_t n_tiles = (n_rows + 63u) >> 6;
    for (uint32_t tbase = (blockIdx.x * 4u + wid) * 64u; tbase < n_tiles; tbase += gridDim.x * 256u) {
        const uint32_t tile = tbase + lane;
        const uint32_t tm = tile < n_tiles ? tmax[tile] : 0u;
        unsigned long long hits = __ballot(cut ? (tm > t_key) : (tm >= t_min));
        while (hits) {
            const uint32_t j = (uint32_t)__ffsll((long long)hits) - 1u;
            hits &= hits - 1ull;
            const uint32_t i = (tbase + j) * 64u + lane;
            const uint32_t key = i < n_rows ? keys[i] : 0u;
            const bool acc = cut ? (key > t_key) : (key >= t_min);
            const unsigned long long m = __ballot(acc);
            if (m) {
                uint32_t pos0 = 0;
                if (lane == 0) pos0 = atomicAdd(&ctl->cand_count, (uint32_t)__popcll(m));
                pos0 = __shfl(pos0, 0);
                if (acc) {
                    const uint32_t pos = pos0 + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    if (pos < CAND_CAP) cand[pos] = i;
                }
            }
        }
    }
    if (cut && blockIdx.x == 0) {
        if (tid == 0) taken_sh = 0;
        __syncthreads();
        for (uint32_t base = 0; base < n_rows; base += 256u) {
            const uint32_t taken = taken_sh;
            if (taken >= quota) break;
            const uint32_t i = base + tid;
            const bool tie = i < n_rows && keys[i] == t_key;
            const unsigned long long m = __ballot(tie);
            if (lane == 0) wcnt[wid] = (uint32_t)__popcll(m);
            __syncthreads();
            uint32_t off = taken;
            for (int w = 0; w < wid; ++w) off += wcnt[w];
            if (tie) {
                const uint32_t pos = off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                if (pos < quota) cand[above3 + pos] = i;
            }
            __syncthreads();
            if (tid == 0) taken_sh = taken + wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------
// final: (optionally) re-score every candidate with the reference arithmetic,
// sort by (score key, row) in one workgroup, certify, write the top k.
// ---------------------------------------------------------------------------
template <int METRIC>
__device__ float exact_score_row(const float* __restrict__ v, const float* __restrict__ q,
                                 uint32_t d, float na) {
    float acc0 = 0.0f, acc1 = 0.0f;
    if ((d & 3u) == 0u) {
        const v4f* v4 = reinterpret_cast<const v4f*>(v);
        for (uint32_t j = 0; j < d / 4; ++j) {
            const v4f x = v4[j];
            accum<METRIC>(q[j * 4 + 0], x.x, acc0, acc1);
            accum<METRIC>(q[j * 4 + 1], x.y, acc0, acc1);
            accum<METRIC>(q[j * 4 + 2], x.z, acc0, acc1);
            accum<METRIC>(q[j * 4 + 3], x.w, acc0, acc1);
        }
    } else {
        for (uint32_t j = 0; j < d; ++j) accum<METRIC>(q[j], v[j], acc0, acc1);
    }
    return finish_score<METRIC>(acc0, acc1, na);
}


// Exact re-score of a batch's candidate rows with the scan's staging: a wave takes 64
// candidates, brings their rows in as 64 x 32 tiles through its private LDS image
// (coalesced 16-byte loads along each row) and every lane sums its own row left to
// right. One thread per row straight from memory thrashes the vector cache (64 rows
// x 3 KB per wave) and took 168 us for 256 queries x ~250 rows of 768 floats.
// grid (queries, CAND_CAP / 256); writes the ordered exact key of candidate i of
// query y to out_keys[y * CAND_CAP + i].
template <int METRIC>
__global__ void __launch_bounds__(256)
rescore_kernel(const float* __restrict__ corpus, const float* __restrict__ Q, uint32_t n_rows, uint32_t d,
               const SearchCtl* __restrict__ ctls, const uint32_t* __restrict__ cand,
               uint32_t* __restrict__ out_keys) {
    constexpr int CW = 32, STRIDE = CW + 4;
    // per wave: 64 rows of a chunk, and the query's 32 components of the same chunk behind them
    __shared__ __attribute__((aligned(16))) float tiles[4][65 * STRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t y = blockIdx.x;
    const SearchCtl* ctl = ctls + y;
    uint32_t C = ctl->cand_count;
    if (C > CAND_CAP) C = CAND_CAP;
    const uint32_t base = blockIdx.y * 256u + (uint32_t)wid * 64u;
    if (base >= C) return;  // tiles are private to a wave: no workgroup barrier below
    const uint32_t* cq = cand + (size_t)y * CAND_CAP;
    const float* __restrict__ q = Q + (size_t)y * d;
    float* tile = tiles[wid];
    const int lrow = lane >> 3, lcol = lane & 7;
    const float* rptr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t ci = min(base + (uint32_t)(i * 8 + lrow), C - 1u);
        rptr[i] = corpus + (size_t)min(cq[ci], n_rows - 1u) * d;
    }
    const uint32_t d4 = d >> 2, n_chunks = (d + CW - 1) / CW;
    // PF chunks of loads in flight: the rows are scattered over the corpus and a wave has nothing else
    // to do while it waits. The query's components travel with them (every lane group of 8 loads the
    // chunk's 128 bytes; they are read back from LDS as broadcasts): fetched through the scalar cache
    // instead, chunk by chunk, they cost a round trip to L2 per 32 columns that nothing overlapped
    // (1000 queries x ~25 rows of 384 floats: 25 us; 256 x ~40 rows of 768: 31 us).
#ifndef RESCORE_PF
#define RESCORE_PF 2
#endif
    constexpr int PF = RESCORE_PF;
    v4f pre[PF][8], pre_q[PF];
    auto issue = [&](uint32_t c, v4f (&dst)[8], v4f& dst_q) {
        const uint32_t col4 = min(c * 8u + (uint32_t)lcol, d4 - 1u);
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[i] = reinterpret_cast<const v4f*>(rptr[i])[col4];
        dst_q = reinterpret_cast<const v4f*>(q)[col4];
    };
#pragma unroll
    for (int p = 0; p < PF; ++p) issue((uint32_t)p, pre[p], pre_q[p]);
    float acc0 = 0.0f, acc1 = 0.0f;
    float* qrow = tile + 64 * STRIDE;
    auto stage = [&](const v4f (&src)[8], const v4f& src_q) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            *reinterpret_cast<v4f*>(&tile[(i * 8 + lrow) * STRIDE + lcol * 4]) = src[i];
        if (lrow == 0) *reinterpret_cast<v4f*>(&qrow[lcol * 4]) = src_q;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    const float* myrow = tile + lane * STRIDE;
    // Groups of PF whole chunks: one chain of basic blocks without a branch, every slot staged, refilled
    // (past the end: a clamped, harmless address) and summed in turn, so that the loads in flight are
    // the same at every step and the compiler waits for a slot with a counted vmcnt.
    const uint32_t n_whole = d / CW;
    uint32_t c0 = 0;
    for (; c0 + PF <= n_whole; c0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            stage(pre[p], pre_q[p]);
            issue(c0 + (uint32_t)p + PF, pre[p], pre_q[p]);
#pragma unroll
            for (int j = 0; j < CW / 4; ++j) {
                const v4f x = *reinterpret_cast<const v4f*>(myrow + j * 4);
                const v4f qv = *reinterpret_cast<const v4f*>(qrow + j * 4);
                accum<METRIC>(qv.x, x.x, acc0, acc1);
                accum<METRIC>(qv.y, x.y, acc0, acc1);
                accum<METRIC>(qv.z, x.z, acc0, acc1);
                accum<METRIC>(qv.w, x.w, acc0, acc1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    // what is left: fewer than PF whole chunks and a last narrow one, already on their way
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        const uint32_t c = c0 + (uint32_t)p;
        if (c < n_chunks) {
            stage(pre[p], pre_q[p]);
            const uint32_t len = min((uint32_t)CW, d - c * CW);
            for (uint32_t j = 0; j < len / 4; ++j) {
                const v4f x = *reinterpret_cast<const v4f*>(myrow + j * 4);
                const v4f qv = *reinterpret_cast<const v4f*>(qrow + j * 4);
                accum<METRIC>(qv.x, x.x, acc0, acc1);
                accum<METRIC>(qv.y, x.y, acc0, acc1);
                accum<METRIC>(qv.z, x.z, acc0, acc1);
                accum<METRIC>(qv.w, x.w, acc0, acc1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    const uint32_t i = base + (uint32_t)lane;
    if (i < C) out_keys[(size_t)y * CAND_CAP + i] = score_key(finish_score<METRIC>(acc0, acc1, ctl->na));
}

// Streaming / batched modes: every row left behind has an approximate score
// <= t (the select threshold) and approximate and exact scores differ by at
// most eps, so the exact top-k is final when its last member clears t + eps.
// key_mode: 0 keys are approximate scores of the streaming pass (IEEE fused multiply-adds,
// any order: d * 2^-24 each for the approximate and the reference sum);
// 1 keys of the f32 matrix-core pass: its products are exact but how the matrix pipe rounds
// its running sums is not documented, so each of its d steps is given 2^-22 (as in the bf16
// bound), 5d * 2^-24 with the reference's own d;
// 2 keys of the f32 matrix-core Euclidean pass (negated squared distance), same steps;
// 3 keys of the bf16 prefilter (Euclidean: negated squared distance);
// 4 keys of the single-query scan of the bf16 mirror (score space for every metric).
template <int METRIC>
__device__ uint32_t certify(const SearchCtl* __restrict__ ctl, uint32_t d, float na, uint32_t t_key,
                            float ek, int key_mode = 0) {
    uint32_t status = ST_OK;
    const float u = 5.9604645e-08f;  // 2^-24
    // (cosine: one more d for the two row norms, each off by d/2 * 2^-24 relative in either path)
    const float steps = ((key_mode == 1 || key_mode == 2) ? 5.0f : 2.0f) + (METRIC == 0 ? 1.0f : 0.0f);
    float eps = (steps * (float)d + 64.0f) * u;
    const float vmax = sqrtf(ctl->vmax_sq);
    const float rho = ctl->rho;
    // the bound assumes ||q||^2 is a normal number well inside the f32 range
    const float nsq = ctl->na_sq;
    if (!(nsq == 0.0f || (nsq >= 0x1p-100f && nsq < 0x1p100f))) status = ST_UNCERTIFIED;
    const float t = t_key >= 0x007FFFFFu ? key_score(t_key) : -INFINITY;
    if (METRIC == 1 && (key_mode == 2 || key_mode == 3)) {
        // Keys order -(||q||^2 + ||v||^2 - 2 q.v), an approximation of the negated SQUARED
        // distance with absolute error <= eps2. Every row left behind has approximate squared
        // distance >= T2 = -t; the exact k-th distance (squared, plus the rounding of the
        // exact path) must stay below T2 minus that error.
        const float span = na + vmax;
        const float eps2 = key_mode == 3 ? bf16_eps<1>(d, na, vmax, rho) : eps * span * span * 1.01f + 1e-30f;
        const float T2 = -t;
        const float dk2 = ek * ek * (1.0f + eps);
        if (!(dk2 + eps2 < T2) || !(eps2 < INFINITY)) status = ST_UNCERTIFIED;
        return status;
    }
    if (key_mode == 3) eps = bf16_eps<METRIC>(d, na, vmax, rho);
    else if (key_mode == 4) eps = mirror_eps<METRIC>(d, na, vmax);
    else if (METRIC == 2) eps = eps * na * vmax * 1.01f + 1e-30f;
    else if (METRIC == 1) eps = eps * (na + vmax) * 1.01f + 1e-20f;
    if (!(ek > t + eps) || !(eps < INFINITY)) status = ST_UNCERTIFIED;
    return status;
}

template <int METRIC, bool RESCORE>
__global__ void __launch_bounds__(1024)
final_kernel(const float* __restrict__ corpus, const float* __restrict__ q, uint32_t n_rows,
             uint32_t d, uint32_t live, SearchCtl* __restrict__ ctl,
             const uint32_t* __restrict__ keys, const uint32_t* __restrict__ cand,
             uint32_t* __restrict__ out_rows, float* __restrict__ out_scores,
             uint32_t* __restrict__ out_n, uint32_t* __restrict__ out_status,
             size_t keys_stride, uint32_t out_stride, int use_count,
             const uint32_t* __restrict__ pre_keys = nullptr) {
    __shared__ unsigned long long sk[CAND_CAP];
    const int tid = threadIdx.x;
    {   // blockIdx.x = query of a batch (a single query launches one workgroup)
        const uint32_t y = blockIdx.x;
        q += (size_t)y * d;
        ctl += y;
        keys += (size_t)y * keys_stride;
        cand += (size_t)y * CAND_CAP;
        if (pre_keys) pre_keys += (size_t)y * CAND_CAP;  // exact keys from rescore_kernel, by candidate
        out_rows += (size_t)y * out_stride;
        out_scores += (size_t)y * out_stride;
        out_n += y;
        if (out_status) out_status += y;
    }
    // use_count != 0: the candidates were gathered against a given threshold and
    // their number is only known from the append cursor; 2 and 3 also name the
    // key space of the pass that gathered them (certify's key_mode)
    uint32_t C = use_count ? ctl->cand_count : ctl->expect;
    const bool overflow = C > CAND_CAP;
    if (C > CAND_CAP) C = CAND_CAP;
    const float na = ctl->na;
    if (C <= 64u) {
        // the usual case behind the bf16 prefilter's prune stage (some tens of rows): one wave sorts
        // them in registers, without a barrier; the other waves leave at once
        if (tid >= 64) return;
        const int lane = tid;
        unsigned long long v = 0ull;
        if ((uint32_t)lane < C) {
            const uint32_t row = cand[lane];
            uint32_t skey;
            if (RESCORE)
                skey = pre_keys ? pre_keys[lane]
                                : score_key(exact_score_row<METRIC>(corpus + (size_t)row * d, q, d, na));
            else
                skey = keys[row];
            v = rank_key(skey, row);
        }
#pragma unroll
        for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, stride);
                const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), stride);
                const unsigned long long o = ((unsigned long long)hi << 32) | lo;
                const bool desc = (lane & size) == 0;
                const bool lower = (lane & stride) == 0;
                const unsigned long long mx = v > o ? v : o, mn = v > o ? o : v;
                v = (lower == desc) ? mx : mn;
            }
        }
        const uint32_t k = ctl->k;
        const uint32_t n_out = k < C ? k : C;
        const float sc = key_score((uint32_t)(v >> 32));
        if ((uint32_t)lane < n_out) {
            out_rows[lane] = 0xFFFFFFFFu - (uint32_t)(v & 0xFFFFFFFFull);
            out_scores[lane] = (METRIC == 1 && sc == 0.0f) ? -0.0f : sc;  // -(sqrt(0)) in the reference
        }
        const float ek = __shfl(sc, n_out > 0 ? (int)n_out - 1 : 0);
        if (lane == 0) {
            *out_n = n_out;
            uint32_t status = ST_OK;
            if (RESCORE && C < live && n_out > 0)
                status = certify<METRIC>(ctl, d, na, ctl->t_key, ek, use_count);
            ctl->levels = ctl->resolved ? ctl->resolved : 3;
            ctl->status = status;
            if (out_status) *out_status = status;
        }
        return;
    }
    uint32_t P = 64;
    while (P < C) P <<= 1;
    for (uint32_t i = tid; i < P; i += 1024u) {
        unsigned long long v = 0ull;
        if (i < C) {
            const uint32_t row = cand[i];
            uint32_t skey;
            if (RESCORE)
                skey = pre_keys ? pre_keys[i]
                                : score_key(exact_score_row<METRIC>(corpus + (size_t)row * d, q, d, na));
            else
                skey = keys[row];
            v = rank_key(skey, row);
        }
        sk[i] = v;
    }
    // bitonic sort, descending
    for (uint32_t size = 2; size <= P; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (uint32_t i = tid; i < P / 2; i += 1024u) {
                const uint32_t pos = 2 * i - (i & (stride - 1));
                const unsigned long long a = sk[pos], b = sk[pos + stride];
                const bool desc = (pos & size) == 0;
                if (desc ? (a < b) : (a > b)) { sk[pos] = b; sk[pos + stride] = a; }
            }
        }
    }
    __syncthreads();
    const uint32_t k = ctl->k;
    const uint32_t n_out = k < C ? k : C;
    for (uint32_t i = tid; i < n_out; i += 1024u) {
        const unsigned long long v = sk[i];
        out_rows[i] = 0xFFFFFFFFu - (uint32_t)(v & 0xFFFFFFFFull);
        float s = key_score((uint32_t)(v >> 32));
        if (METRIC == 1 && s == 0.0f) s = -0.0f;  // -(sqrt(0)) in the reference
        out_scores[i] = s;
    }
    if (tid == 0) {
        *out_n = n_out;
        uint32_t status = ST_OK;
        if (RESCORE && C < live && n_out > 0)
            status = certify<METRIC>(ctl, d, na, ctl->t_key, key_score((uint32_t)(sk[n_out - 1] >> 32)),
                                     use_count);  // 0 streaming, 1 / 2 f32 matrix cores, 3 bf16 prefilter
        if (overflow) status = ST_OVERFLOW;
        ctl->levels = ctl->resolved ? ctl->resolved : 3;
        ctl->status = status;
        if (out_status) *out_status = status;
    }
}


// ---------------------------------------------------------------------------
// Candidate gather of the one-workgroup select stages (1024 threads): append every
// row with lo <= key < hi to sk[] (as its rank key, or as its bare row index), counting
// in *n_cand. A scan leaves three levels to descend: the best key of each scan wave
// (wmax), of each 64-row tile (tmax; tile t was scanned by wave t % n_waves), and the
// keys. Each level is one flat, fully parallel sweep -- every thread issues its
// independent loads back to back, so a level costs about one memory round trip:
//   waves that reach lo -> hit[];  their tiles that reach lo -> a tile list;  the 64
//   keys of each listed tile.
// The tile list lives in the upper half of sk[] (as 32-bit words), so this form needs
// the candidates to stay within the lower half: callers use it when at most
// CAND_CAP / 2 rows can qualify, and the wave-by-wave form below otherwise.
// All 1024 threads must call it; ends with a barrier.
// (Asking for ALL tile maxima at the kernel's start, beside the histogram -- 20 per thread at 1.25 M
// rows -- and so skipping the level of the wave maxima measured no gain: 17.2 against 16.9 us.)
// ---------------------------------------------------------------------------
template <int NT /* threads of the workgroup */>
__device__ __forceinline__ void gather_flat(uint32_t lo, uint32_t hi, bool rows_only, uint32_t n_rows,
                                            uint32_t n_tiles, uint32_t n_waves, uint32_t my_wmax,
                                            const uint32_t* __restrict__ wmax, const uint32_t* __restrict__ tmax,
                                            const uint32_t* __restrict__ keys, unsigned long long* sk,
                                            uint32_t* hit, uint32_t* n_hit, uint32_t* n_thit, uint32_t* n_cand) {
    const uint32_t tid = threadIdx.x;
    uint32_t* thit = reinterpret_cast<uint32_t*>(sk + CAND_CAP / 2);  // CAND_CAP words
    for (uint32_t w0 = 0; w0 < n_waves; w0 += (uint32_t)NT) {
        const uint32_t w = w0 + tid;
        const uint32_t wm = w0 == 0 ? my_wmax : ((w < n_waves && w < n_tiles) ? wmax[w] : 0u);
        if (wm >= lo) {
            const uint32_t pos = atomicAdd(n_hit, 1u);
            if (pos < CAND_CAP) hit[pos] = w;
        }
    }
    __syncthreads();
    const uint32_t nh = *n_hit < CAND_CAP ? *n_hit : CAND_CAP;
    const uint32_t tpw = (n_tiles + n_waves - 1u) / n_waves;  // tiles per scan wave
    const uint32_t total = nh * tpw;
    for (uint32_t i0 = 0; i0 < total; i0 += 4u * (uint32_t)NT) {
        uint32_t tile[4], tm[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t i = i0 + tid + (uint32_t)NT * e;
            const uint32_t h = i / tpw, j = i - h * tpw;
            tile[e] = i < total ? hit[h] + j * n_waves : 0xFFFFFFFFu;
            tm[e] = tile[e] < n_tiles ? tmax[tile[e]] : 0u;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (tm[e] >= lo) {
                const uint32_t pos = atomicAdd(n_thit, 1u);
                if (pos < CAND_CAP) thit[pos] = tile[e];
            }
        }
    }
    __syncthreads();
    const uint32_t nt = *n_thit < CAND_CAP ? *n_thit : CAND_CAP;
    const uint32_t lane = tid & 63u;
    for (uint32_t t0 = 0; t0 < nt; t0 += (uint32_t)NT / 16u) {  // NT / 64 waves x 4 tiles per round
        uint32_t row[4], key[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t ti = t0 + (tid >> 6) * 4u + e;
            row[e] = ti < nt ? thit[ti] * 64u + lane : 0xFFFFFFFFu;
            key[e] = row[e] < n_rows ? keys[row[e]] : 0u;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (key[e] >= lo && key[e] < hi) {
                const uint32_t pos = atomicAdd(n_cand, 1u);
                if (pos < CAND_CAP / 2) sk[pos] = rows_only ? (unsigned long long)row[e] : rank_key(key[e], row[e]);
            }
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------
// select_final: the common case in ONE workgroup. The scan's histogram gives
// the coarse bin that holds the need-th best key; when the rows at or above
// that bin fit the sort they are collected through the two levels of tile
// maxima (supertile = 64 tiles = 4096 rows), (optionally re-scored,) sorted and
// written out; the workgroup then zeroes what the next query's scan
// accumulates into. When the bin is too crowded (clustered scores, big tie
// groups) the status ST_NEED_REFINE asks the host for the multi-level pipeline.
// ---------------------------------------------------------------------------
// The body, for a workgroup of NT threads (1024: the stand-alone kernel; 256: the workgroup that a fused
// scan launch sets aside for the PREVIOUS query's select). LDS comes from the caller: sk[CAND_CAP],
// hit[CAND_CAP], sh[24], cnt[4].
template <int METRIC, bool RESCORE, int NT>
__device__ __forceinline__ void
select_final_body(const float* __restrict__ corpus, const float* __restrict__ q, uint32_t n_rows,
                  uint32_t d, uint32_t live, uint32_t need, uint32_t k, uint32_t n_waves,
                  SearchCtl* __restrict__ ctl, uint32_t* __restrict__ hist,
                  const uint32_t* __restrict__ keys, const uint32_t* __restrict__ tmax,
                  const uint32_t* __restrict__ wmax, uint32_t* __restrict__ out_rows,
                  float* __restrict__ out_scores, uint32_t* __restrict__ out_n,
                  uint32_t* __restrict__ out_status, uint32_t* __restrict__ events,
                  const unsigned long long* __restrict__ wbest, uint32_t epoch,
                  unsigned long long* sk, uint32_t* hit, uint32_t* sh, uint32_t* cnt) {
    uint32_t& n_hit_sh = cnt[0];
    uint32_t& n_cand_sh = cnt[1];
    uint32_t& n_thit_sh = cnt[2];
    uint32_t& keep_sh = cnt[3];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t n_tiles = (n_rows + 63u) >> 6;
    // independent of the threshold: this thread's share of the per-wave maxima
    const uint32_t my_wmax = ((uint32_t)tid < n_waves && (uint32_t)tid < n_tiles) ? wmax[tid] : 0u;   // (waves 0 .. NT - 1)
    float na = 0.0f;
    if (RESCORE) na = ctl->na;
    // A scan with sparse key stores (trag_scan_common.h; single exact queries only) kept the keys
    // of the tiles that reach the need-th largest of the keys its waves published; at least
    // `need` rows lie at or above that key, so the cut starts there and skipped tiles are never read.
    if (wid == 0) {
        uint32_t keep = 0u;
        if (epoch != 0u && wbest) {
            unsigned long long e[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) e[i] = wbest[i * 64 + lane];
            keep = sparse_poll_value(e, epoch, need);
        }
        if (lane == 0) keep_sh = keep;
    }
    uint32_t b1, above1;
    uint32_t inbin1;
    find_bin<NT>(hist, need, b1, above1, sh, &inbin1);
    const uint32_t expect = above1 + inbin1;
    uint32_t status = ST_OK, n_out = 0, C = 0;
    // A crowded coarse bin (clustered scores: Euclidean distances between normalised vectors, big tie
    // groups) holds more rows than the sort. When the scan ran with sparse key stores its bound does
    // the cutting instead: `need` rows reach keep_sh, the rows at or above it are few (about
    // need x rows / rows seen before the waves' last publication) and every one of them has its key
    // stored -- the crowd below is never looked at. Only when that set overflows as well (a tie group
    // of thousands of rows at the very top) is the multi-level form asked for.
    const bool crowded = expect > CAND_CAP;
    const bool by_bound = crowded && keep_sh > (b1 << 21);   // (find_bin's barriers ordered keep_sh)
    if (crowded && !by_bound) {
        status = ST_NEED_REFINE;
    } else {
        if (tid == 0) { n_hit_sh = 0; n_cand_sh = 0; n_thit_sh = 0; }
        __syncthreads();
        const uint32_t t_key = max(b1 << 21, keep_sh);
        const uint32_t t_min = t_key > 1u ? t_key : 1u;  // key 0 marks removed rows
        if (expect <= CAND_CAP / 2 || by_bound) {
            gather_flat<NT>(t_min, 0xFFFFFFFFu, RESCORE, n_rows, n_tiles, n_waves, my_wmax, wmax, tmax, keys, sk, hit,
                        &n_hit_sh, &n_thit_sh, &n_cand_sh);
        } else {
        // (more rows than the flat form holds) scan waves whose best key reaches the threshold ...
        for (uint32_t w0 = 0; w0 < n_waves; w0 += (uint32_t)NT) {
            const uint32_t w = w0 + tid;
            const uint32_t wm = w0 == 0 ? my_wmax : ((w < n_waves && w < n_tiles) ? wmax[w] : 0u);
            if (wm >= t_min) {
                const uint32_t pos = atomicAdd(&n_hit_sh, 1u);
                if (pos < CAND_CAP) hit[pos] = w;
            }
        }
        __syncthreads();
        const uint32_t nh = n_hit_sh < CAND_CAP ? n_hit_sh : CAND_CAP;
        // ... then their tiles (w, w + n_waves, ...), then the rows of the tiles that pass
        for (uint32_t h = wid; h < nh; h += (uint32_t)NT / 64u) {
            const uint32_t w = hit[h];
            for (uint32_t tb = 0; w + tb * n_waves < n_tiles; tb += 64u) {
                const uint32_t tile = w + (tb + lane) * n_waves;
                const uint32_t tm = tile < n_tiles ? tmax[tile] : 0u;
                unsigned long long hits = __ballot(tm >= t_min);
                while (hits) {
                    const uint32_t j = (uint32_t)__ffsll((long long)hits) - 1u;
                    hits &= hits - 1ull;
                    const uint32_t i = (w + (tb + j) * n_waves) * 64u + lane;
                    const uint32_t key = i < n_rows ? keys[i] : 0u;
                    const bool acc = key >= t_min;
                    const unsigned long long m = __ballot(acc);
                    if (m) {
                        uint32_t pos0 = 0;
                        if (lane == 0) pos0 = atomicAdd(&n_cand_sh, (uint32_t)__popcll(m));
                        pos0 = __shfl(pos0, 0);
                        if (acc) {
                            const uint32_t pos = pos0 + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                            if (pos < CAND_CAP) sk[pos] = RESCORE ? (unsigned long long)i : rank_key(key, i);
                        }
                    }
                }
            }
        }
        __syncthreads();
        }
        C = n_cand_sh < CAND_CAP ? n_cand_sh : CAND_CAP;
        n_out = k < C ? k : C;
        // (cut by the bound: the flat gather holds CAND_CAP / 2 rows, CAND_CAP waves and tiles)
        const bool overflow = by_bound && (n_cand_sh > CAND_CAP / 2 || n_hit_sh > CAND_CAP || n_thit_sh > CAND_CAP);
        if (overflow) { status = ST_NEED_REFINE; C = 0; n_out = 0; }
        float ek = 0.0f;  // score of the last returned row (thread 0)
        if (overflow) {
        } else if (C <= 64u) {
            // the usual case: one wave sorts its candidates in registers
            if (wid == 0) {
                unsigned long long v = 0ull;
                if ((uint32_t)lane < C) {
                    v = sk[lane];
                    if (RESCORE) {
                        const uint32_t row = (uint32_t)v;
                        v = rank_key(score_key(exact_score_row<METRIC>(corpus + (size_t)row * d, q, d, na)), row);
                    }
                }
#pragma unroll
                for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
                    for (int stride = size >> 1; stride > 0; stride >>= 1) {
                        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, stride);
                        const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), stride);
                        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
                        const bool desc = (lane & size) == 0;
                        const bool lower = (lane & stride) == 0;
                        const unsigned long long mx = v > o ? v : o, mn = v > o ? o : v;
                        v = (lower == desc) ? mx : mn;
                    }
                }
                float sc = key_score((uint32_t)(v >> 32));
                if (METRIC == 1 && sc == 0.0f) sc = -0.0f;  // -(sqrt(0)) in the reference
                if ((uint32_t)lane < n_out) {
                    out_rows[lane] = 0xFFFFFFFFu - (uint32_t)(v & 0xFFFFFFFFull);
                    out_scores[lane] = sc;
                }
                ek = __shfl(key_score((uint32_t)(v >> 32)), n_out > 0 ? (int)n_out - 1 : 0);
            }
        } else if (C <= 256u) {
            // up to 256 candidates (the usual case on a corpus of a million rows or more, where the
            // coarse bin of the need-th key holds some tens of rows): rank by counting -- element
            // tid % 256 against a quarter of the others per thread, two barriers -- instead of the 28
            // barriers of a 128-element sorting network (6.9 us of this kernel at 1 M rows)
            uint32_t* rank = hit;   // (the wave list is no longer needed)
            constexpr uint32_t PARTS = NT / 256, SPAN = 256u / PARTS;
            const uint32_t i = (uint32_t)tid & 255u, part = (uint32_t)tid >> 8;
            if (RESCORE) {
                __syncthreads();
                if (part == 0u && i < C) {
                    const uint32_t row = (uint32_t)sk[i];
                    sk[i] = rank_key(score_key(exact_score_row<METRIC>(corpus + (size_t)row * d, q, d, na)), row);
                }
            }
            if (part == 0u) rank[i] = 0u;
            __syncthreads();
            const unsigned long long mine = i < C ? sk[i] : 0ull;
            uint32_t above = 0u;
            const uint32_t j0 = part * SPAN, j1 = min(j0 + SPAN, C);
            for (uint32_t j = j0; j < j1; ++j) above += sk[j] > mine ? 1u : 0u;
            if (above) atomicAdd(&rank[i], above);
            __syncthreads();
            if (part == 0u && i < C) {
                const uint32_t r = rank[i];
                if (r < n_out) {
                    float sc = key_score((uint32_t)(mine >> 32));
                    if (r == n_out - 1u) sh[20] = __float_as_uint(sc);
                    if (METRIC == 1 && sc == 0.0f) sc = -0.0f;
                    out_rows[r] = 0xFFFFFFFFu - (uint32_t)(mine & 0xFFFFFFFFull);
                    out_scores[r] = sc;
                }
            }
            __syncthreads();
            if (n_out > 0) ek = __uint_as_float(sh[20]);
        } else {
            uint32_t P = 128;
            while (P < C) P <<= 1;
            for (uint32_t i = tid; i < P; i += (uint32_t)NT) {
                unsigned long long v = 0ull;
                if (i < C) {
                    v = sk[i];
                    if (RESCORE) {
                        const uint32_t row = (uint32_t)v;
                        v = rank_key(score_key(exact_score_row<METRIC>(corpus + (size_t)row * d, q, d, na)), row);
                    }
                }
                sk[i] = v;
            }
            for (uint32_t size = 2; size <= P; size <<= 1) {
                for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
                    __syncthreads();
                    for (uint32_t i = tid; i < P / 2; i += (uint32_t)NT) {
                        const uint32_t pos = 2 * i - (i & (stride - 1));
                        const unsigned long long a = sk[pos], b = sk[pos + stride];
                        const bool desc = (pos & size) == 0;
                        if (desc ? (a < b) : (a > b)) { sk[pos] = b; sk[pos + stride] = a; }
                    }
                }
            }
            __syncthreads();
            for (uint32_t i = tid; i < n_out; i += (uint32_t)NT) {
                const unsigned long long v = sk[i];
                out_rows[i] = 0xFFFFFFFFu - (uint32_t)(v & 0xFFFFFFFFull);
                float sc = key_score((uint32_t)(v >> 32));
                if (METRIC == 1 && sc == 0.0f) sc = -0.0f;
                out_scores[i] = sc;
            }
            if (n_out > 0) ek = key_score((uint32_t)(sk[n_out - 1] >> 32));
        }
        if (RESCORE && tid == 0 && C < live && n_out > 0) status = certify<METRIC>(ctl, d, na, t_key, ek);
    }
    // leave the accumulator clean for the next query's scan
    for (uint32_t i = tid; i < HIST_BINS; i += (uint32_t)NT) hist[i] = 0u;
    if (tid == 0) {
        *out_n = n_out;
        ctl->expect = C;
        ctl->t_key = b1 << 21;
        ctl->levels = 1;
        ctl->keep_key = keep_sh;
        ctl->status = status;
        if (out_status) *out_status = status;
        if (status == ST_NEED_REFINE && events)
            __hip_atomic_fetch_add(events, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}


template <int METRIC, bool RESCORE>
__global__ void __launch_bounds__(1024)
select_final_kernel(const float* __restrict__ corpus, const float* __restrict__ q, uint32_t n_rows,
                    uint32_t d, uint32_t live, uint32_t need, uint32_t k, uint32_t n_waves,
                    SearchCtl* __restrict__ ctl, uint32_t* __restrict__ hist,
                    const uint32_t* __restrict__ keys, const uint32_t* __restrict__ tmax,
                    const uint32_t* __restrict__ wmax, uint32_t* __restrict__ out_rows,
                    float* __restrict__ out_scores, uint32_t* __restrict__ out_n,
                    uint32_t* __restrict__ out_status, MultiStride ms, uint32_t out_stride,
                    uint32_t* __restrict__ events, const unsigned long long* __restrict__ wbest,
                    uint32_t epoch) {
    // Every argument is asked for here, at the top: left alone, the compiler fetches each group of
    // arguments just before its first use, and this kernel -- one workgroup, nothing to overlap with,
    // behind a scan that left neither the argument block nor the code in any cache -- then pays a
    // round trip to memory for each group in turn (four of them before the histogram is even read:
    // 7.8 of the kernel's 19.6 us at 1 M rows).
    asm volatile("" ::"s"(corpus), "s"(q), "s"(n_rows), "s"(d), "s"(live), "s"(need), "s"(k), "s"(n_waves), "s"(ctl),
                 "s"(hist), "s"(keys), "s"(tmax), "s"(wmax));
    asm volatile("" ::"s"(out_rows), "s"(out_scores), "s"(out_n), "s"(out_status), "s"(ms.zero_bytes), "s"(ms.keys),
                 "s"(ms.tmax), "s"(ms.wmax), "s"(out_stride), "s"(events), "s"(wbest), "s"(epoch));
    // one workgroup per query of a multi-query scan (blockIdx.x = query; strides 0 for one query)
    {
        const size_t b = blockIdx.x;
        q += b * d;
        ctl = reinterpret_cast<SearchCtl*>(reinterpret_cast<char*>(ctl) + b * ms.zero_bytes);
        hist = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(hist) + b * ms.zero_bytes);
        keys += b * ms.keys;
        tmax += b * ms.tmax;
        wmax += b * ms.wmax;
        out_rows += b * out_stride;
        out_scores += b * out_stride;
        out_n += b;
        if (out_status) out_status += b;
    }
    __shared__ unsigned long long sk[CAND_CAP];
    __shared__ uint32_t hit[CAND_CAP];
    __shared__ uint32_t sh[24];
    __shared__ uint32_t cnt[4];
    select_final_body<METRIC, RESCORE, 1024>(corpus, q, n_rows, d, live, need, k, n_waves, ctl, hist, keys, tmax, wmax,
                                             out_rows, out_scores, out_n, out_status, events, wbest, epoch, sk, hit, sh,
                                             cnt);
}

// ---------------------------------------------------------------------------
// select_mirror: the one-workgroup stage after a scan of the bf16 mirror.
//
// keys[] hold APPROXIMATE scores, each within eps = mirror_eps of the reference's
// f32 score (rows outside the bound's range carry +inf). The workgroup
//   1. finds the coarse bin of the need-th best key (the cut); the rows at or above
//      it are the first candidates (collected through wave and tile maxima, as
//      select_final does);
//   2. takes A = the k-th best approximate score among the candidates with a valid
//      bound. k rows have an exact score >= A - eps, so every row of the exact top k
//      has an approximate score >= T = A - 2 eps; when T lies below the cut the rows
//      between T and the cut are collected too;
//   3. the rows with approximate score >= T (plus the +inf rows) go to cand[]; they
//      are scored from the f32 corpus in the reference's order by rescore_wave_kernel
//      and ordered by mirror_final_kernel.
// Crowded bins (ST_NEED_REFINE) send the query to the multi-level form below; a list
// beyond its capacity or a query outside the bound's range is repeated with the exact scan.
// ---------------------------------------------------------------------------
// Mirror scan on a corpus whose scores crowd the coarse bins (the multi-level select is
// the plan): after the radix levels, fix the need-th best approximate key A (need = k; an
// early-resolved level gives a lower bound of it, which only admits more rows) and move the
// threshold down to T = A - 2 eps: every row of the exact top k has an approximate score
// >= T. compact_kernel then appends the rows at or above T, rescore_kernel / final_kernel
// score them exactly. One workgroup of 256 threads.
template <int METRIC>
__global__ void __launch_bounds__(256)
mirror_threshold_kernel(SearchCtl* __restrict__ ctl, const uint32_t* __restrict__ hist, uint32_t d) {
    __shared__ uint32_t sh[8];
    uint32_t t_key;
    if (ctl->resolved) {
        t_key = ctl->t_key;
    } else {
        uint32_t b3, a3;
        find_bin<256>(hist + 2 * HIST_BINS, ctl->need - ctl->above2, b3, a3, sh);
        t_key = (ctl->b1 << 21) | (ctl->b2 << 10) | b3;
    }
    if (threadIdx.x == 0) {
        const float na = ctl->na, vmax = sqrtf(ctl->vmax_sq), nsq = ctl->na_sq;
        const float eps = mirror_eps<METRIC>(d, na, vmax);
        uint32_t T_key = 1u;  // outside the bound's range: every row is a candidate (overflows, repeated exactly)
        if ((nsq == 0.0f || (nsq >= 0x1p-100f && nsq < 0x1p100f)) && eps < INFINITY && t_key >= 0x007FFFFFu) {
            const float T = key_score(t_key) - 2.02f * eps;
            if (T == T && T > -INFINITY) T_key = score_key(T);
        }
        ctl->t_key = T_key;
        ctl->resolved = 3;
        ctl->cut_ties = 0;
        ctl->cand_count = 0;
        ctl->status = ST_OK;
    }
}

constexpr uint32_t MIRROR_RESCORE = 2048;  // most rows scored exactly

// Exact score of one row computed by a whole wave: lane l forms the product of
// element 64c + l (rounded once, as the reference's x * y), then the products are
// added strictly left to right through v_readlane, so the sum has the reference's
// bits. Every lane returns the same value. All 64 lanes must call it.
template <int METRIC>
__device__ float exact_score_wave(const float* __restrict__ v, const float* __restrict__ q, uint32_t d,
                                  float na, int lane) {
    const uint32_t n_chunks = (d + 63u) / 64u;
    auto load = [&](const float* __restrict__ p, uint32_t c) -> float {
        const uint32_t j = c * 64u + (uint32_t)lane;
        return (c < n_chunks && j < d) ? p[j] : 0.0f;
    };
    float acc0 = 0.0f, acc1 = 0.0f;
    float x0 = load(v, 0), q0 = load(q, 0), x1 = load(v, 1), q1 = load(q, 1);
    for (uint32_t c = 0; c < n_chunks; ++c) {
        const float x2 = load(v, c + 2), q2 = load(q, c + 2);
        float p, p2 = 0.0f;
        if (METRIC == 1) {
            const float t = q0 - x0;
            p = t * t;
        } else {
            p = q0 * x0;
            if (METRIC == 0) p2 = x0 * x0;
        }
        const int b0 = __float_as_int(p), b1 = __float_as_int(p2);
        const uint32_t len = min(64u, d - c * 64u);
        if (len == 64u) {
#pragma unroll
            for (int i = 0; i < 64; ++i) {
                acc0 = acc0 + __int_as_float(__builtin_amdgcn_readlane(b0, i));
                if (METRIC == 0) acc1 = acc1 + __int_as_float(__builtin_amdgcn_readlane(b1, i));
            }
        } else {
            for (uint32_t i = 0; i < len; ++i) {
                acc0 = acc0 + __int_as_float(__builtin_amdgcn_readlane(b0, (int)i));
                if (METRIC == 0) acc1 = acc1 + __int_as_float(__builtin_amdgcn_readlane(b1, (int)i));
            }
        }
        x0 = x1; q0 = q1; x1 = x2; q1 = q2;
    }
    return finish_score<METRIC>(acc0, acc1, na);
}

// One-workgroup prune after a mirror scan (see the header of this section): writes the
// rows to score exactly to cand[], their number to ctl->cand_count, the lowered
// threshold T to ctl->t_key and its verdict to ctl->status; zeroes the coarse histogram.
template <int METRIC>
__global__ void __launch_bounds__(1024)
select_mirror_kernel(uint32_t n_rows, uint32_t d, uint32_t live, uint32_t need, uint32_t k, uint32_t n_waves,
                     SearchCtl* __restrict__ ctl, uint32_t* __restrict__ hist,
                     const uint32_t* __restrict__ keys, const uint32_t* __restrict__ tmax,
                     const uint32_t* __restrict__ wmax, uint32_t* __restrict__ cand,
                     uint32_t* __restrict__ events, const unsigned long long* __restrict__ wbest,
                     uint32_t epoch) {
    __shared__ unsigned long long sk[CAND_CAP];
    __shared__ uint32_t hit[CAND_CAP];
    __shared__ uint32_t sh[24];
    __shared__ uint32_t n_hit_sh, n_cand_sh, n_thit_sh, extra_sh, nf_sh, ns_sh, akey_sh, keep_sh;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t n_tiles = (n_rows + 63u) >> 6;
    const uint32_t my_wmax = ((uint32_t)tid < n_waves && (uint32_t)tid < n_tiles) ? wmax[tid] : 0u;
    // (the control block's fields are fetched up front: their latency hides behind the histogram scan)
    const float na = ctl->na, nsq = ctl->na_sq, vmax = sqrtf(ctl->vmax_sq);
    uint32_t b1, above1, inbin1;
    // A scan with sparse key stores (trag_scan_common.h) kept the keys of the tiles that reach
    // the bound below, which this stage recomputes from the waves' published keys: at least k
    // bounded rows lie at or above it and T never falls under it, so the cut starts there.
    if (wid == 0) {
        uint32_t keep = 0u;
        if (epoch != 0u && wbest) {
            unsigned long long e[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) e[i] = wbest[i * 64 + lane];
            const float eps0 = mirror_eps<METRIC>(d, na, vmax);
            if ((nsq == 0.0f || (nsq >= 0x1p-100f && nsq < 0x1p100f)) && eps0 < INFINITY)
                keep = sparse_keep_key(sparse_poll_value(e, epoch, k), 2.02f * eps0 * 1.01f);
        }
        if (lane == 0) keep_sh = keep;
    }
    find_bin<1024>(hist, need, b1, above1, sh, &inbin1);
    const uint32_t expect = above1 + inbin1;  // (an upper bound when the cut is raised to keep_sh)
    if (tid == 0) { n_hit_sh = 0; n_cand_sh = 0; n_thit_sh = 0; extra_sh = 0; nf_sh = 0; ns_sh = 0; akey_sh = 0; }
    __syncthreads();
    // append the rows with lo <= key < hi to sk[] (through the wave maxima, then the tile
    // maxima of the waves that pass, then the keys of the tiles that pass)
    auto gather = [&](uint32_t lo, uint32_t hi, uint32_t at_most) {
        if (at_most <= CAND_CAP / 2) {  // at_most: bound of the candidate count after this call
            gather_flat<1024>(lo, hi, false, n_rows, n_tiles, n_waves, my_wmax, wmax, tmax, keys, sk, hit, &n_hit_sh,
                        &n_thit_sh, &n_cand_sh);
            return;
        }
        for (uint32_t w0 = 0; w0 < n_waves; w0 += 1024u) {
            const uint32_t w = w0 + tid;
            const uint32_t wm = w0 == 0 ? my_wmax : ((w < n_waves && w < n_tiles) ? wmax[w] : 0u);
            if (wm >= lo) {
                const uint32_t pos = atomicAdd(&n_hit_sh, 1u);
                if (pos < CAND_CAP) hit[pos] = w;
            }
        }
        __syncthreads();
        const uint32_t nh = n_hit_sh < CAND_CAP ? n_hit_sh : CAND_CAP;
        for (uint32_t h = wid; h < nh; h += 16u) {
            const uint32_t w = hit[h];
            for (uint32_t tb = 0; w + tb * n_waves < n_tiles; tb += 64u) {
                const uint32_t tile = w + (tb + lane) * n_waves;
                const uint32_t tm = tile < n_tiles ? tmax[tile] : 0u;
                unsigned long long hits = __ballot(tm >= lo);
                while (hits) {
                    const uint32_t j = (uint32_t)__ffsll((long long)hits) - 1u;
                    hits &= hits - 1ull;
                    const uint32_t i = (w + (tb + j) * n_waves) * 64u + lane;
                    const uint32_t key = i < n_rows ? keys[i] : 0u;
                    const bool acc = key >= lo && key < hi;
                    const unsigned long long m = __ballot(acc);
                    if (m) {
                        uint32_t pos0 = 0;
                        if (lane == 0) pos0 = atomicAdd(&n_cand_sh, (uint32_t)__popcll(m));
                        pos0 = __shfl(pos0, 0);
                        if (acc) {
                            const uint32_t pos = pos0 + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                            if (pos < CAND_CAP) sk[pos] = rank_key(key, i);
                        }
                    }
                }
            }
        }
        __syncthreads();
    };
    uint32_t status = ST_OK, nS = 0, T_key = 1u, C = 0;
    if (expect > CAND_CAP) {
        status = ST_NEED_REFINE;
    } else {
        const uint32_t t_key = max(b1 << 21, keep_sh);
        const uint32_t t_min = t_key > 1u ? t_key : 1u;  // key 0 marks removed rows
        const bool all_in = t_min == 1u;                 // every live row is a candidate
        const uint32_t b_top = (t_min + 0x1FFFFFu) >> 21;  // coarse bins below this one lie wholly under the cut
        gather(t_min, 0xFFFFFFFFu, expect);  // (+inf is the largest key a row can carry: 0xFF800000)
        C = n_cand_sh < CAND_CAP ? n_cand_sh : CAND_CAP;
        // The k-th best approximate key by counting ranks (all threads read the same sk[j] at
        // a time: an LDS broadcast); nothing needs sorting here.
        const uint32_t INF_KEY = 0xFF800000u;  // rows without a valid bound: always scored exactly
        {
            unsigned long long mine[4];
            uint32_t rank[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t i = (uint32_t)tid + 1024u * e;
                mine[e] = i < C ? sk[i] : 0ull;
                if (i < C && (uint32_t)(mine[e] >> 32) == INF_KEY) atomicAdd(&nf_sh, 1u);
            }
            const int live_e = C > 3072u ? 4 : (C > 2048u ? 3 : (C > 1024u ? 2 : 1));
            for (uint32_t j = 0; j < C; ++j) {
                const unsigned long long o = sk[j];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e < live_e) rank[e] += o > mine[e] ? 1u : 0u;
            }
            __syncthreads();
            const uint32_t nf = nf_sh;
            if (nf + k <= C && k > 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t i = (uint32_t)tid + 1024u * e;
                    if (i < C && rank[e] == nf + k - 1u) akey_sh = (uint32_t)(mine[e] >> 32);
                }
            }
            __syncthreads();
        }
        const float eps = mirror_eps<METRIC>(d, na, vmax);
        if (!(nsq == 0.0f || (nsq >= 0x1p-100f && nsq < 0x1p100f)) || !(eps < INFINITY)) status = ST_UNCERTIFIED;
        if (akey_sh >= 0x007FFFFFu) {
            const float T = key_score(akey_sh) - 2.02f * eps;
            T_key = (T == T && T > -INFINITY) ? score_key(T) : 1u;
            if (T_key < 1u) T_key = 1u;
        } else if (!all_in) {
            status = ST_UNCERTIFIED;  // fewer than k bounded rows above the cut
        }
        // T below the cut: the rows between T and the cut join the candidates (their keys are
        // below every candidate so far: the k-th best key stands). The coarse histogram bounds
        // their number before anything is gathered.
        if (status == ST_OK && !all_in && T_key < t_min) {
            const uint32_t cut2 = T_key >> 21;
            if (b_top - cut2 > 64u) {
                status = ST_UNCERTIFIED;
            } else {
                if (wid == 0) {
                    const uint32_t bin = b_top - 1u - (uint32_t)lane;
                    uint32_t h = (bin >= cut2 && bin < b_top) ? hist[bin] : 0u;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o);
                    if (lane == 0) { extra_sh = h; n_hit_sh = 0; n_thit_sh = 0; }
                }
                __syncthreads();
                if (C + extra_sh > CAND_CAP) {
                    status = ST_OVERFLOW;
                } else {
                    gather(T_key, t_min, C + extra_sh);
                    C = n_cand_sh < CAND_CAP ? n_cand_sh : CAND_CAP;
                }
            }
        }
        if (status == ST_OK) {
            for (uint32_t i = tid; i < C; i += 1024u) {
                const unsigned long long v = sk[i];
                if ((uint32_t)(v >> 32) >= T_key) {
                    const uint32_t pos = atomicAdd(&ns_sh, 1u);
                    if (pos < CAND_CAP) cand[pos] = 0xFFFFFFFFu - (uint32_t)(v & 0xFFFFFFFFull);
                }
            }
        }
        __syncthreads();
        nS = ns_sh;
        if (nS > MIRROR_RESCORE) { status = ST_OVERFLOW; nS = 0; }
    }
    // leave the accumulator clean for the next query's scan
    for (uint32_t i = tid; i < HIST_BINS; i += 1024u) hist[i] = 0u;
    if (tid == 0) {
        ctl->cand_count = status == ST_OK ? nS : 0u;
        ctl->expect = C;
        ctl->t_key = T_key;
        ctl->levels = 1;
        ctl->keep_key = keep_sh;
        ctl->status = status;
        if (status == ST_NEED_REFINE && events)
            __hip_atomic_fetch_add(events, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Exact scores of the listed rows, one wave per row, spread over the whole GPU (a row of
// 4096 floats takes one wave ~27 us: the sum is sequential by definition).
// out_keys[i] = ordered key of the exact score of row cand[i], i < min(ctl->cand_count, CAND_CAP).
template <int METRIC>
__global__ void __launch_bounds__(256)
rescore_wave_kernel(const float* __restrict__ corpus, const float* __restrict__ q, uint32_t n_rows, uint32_t d,
                    const SearchCtl* __restrict__ ctl, const uint32_t* __restrict__ cand,
                    uint32_t* __restrict__ out_keys) {
    const int lane = threadIdx.x & 63;
    uint32_t C = ctl->cand_count;
    if (C > CAND_CAP) C = CAND_CAP;
    const float na = ctl->na;
    const uint32_t n_w = gridDim.x * 4u;
    for (uint32_t i = blockIdx.x * 4u + (threadIdx.x >> 6); i < C; i += n_w) {
        const uint32_t row = min(cand[i], n_rows - 1u);
        const float sc = exact_score_wave<METRIC>(corpus + (size_t)row * d, q, d, na, lane);
        if (lane == 0) out_keys[i] = score_key(sc);
    }
}

// Last stage after a mirror scan: order the exactly scored rows (rank by counting), write
// the best k, confirm that the k-th clears the threshold by the bound (true by construction
// when the bound holds), and reset what the next query accumulates into.
template <int METRIC>
__global__ void __launch_bounds__(1024)
mirror_final_kernel(uint32_t d, uint32_t live, SearchCtl* __restrict__ ctl, const uint32_t* __restrict__ cand,
                    const uint32_t* __restrict__ ckeys, uint32_t* __restrict__ out_rows,
                    float* __restrict__ out_scores, uint32_t* __restrict__ out_n,
                    uint32_t* __restrict__ out_status) {
    __shared__ unsigned long long sk[CAND_CAP];
    __shared__ float ek_sh;
    const int tid = threadIdx.x;
    uint32_t status = ctl->status;
    uint32_t C = ctl->cand_count;
    if (C > CAND_CAP) { status = ST_OVERFLOW; C = 0; }
    if (status != ST_OK) C = 0;
    const uint32_t k = ctl->k;
    const uint32_t n_out = k < C ? k : C;
    unsigned long long mine[4];
    uint32_t rank[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint32_t i = (uint32_t)tid + 1024u * e;
        mine[e] = i < C ? rank_key(ckeys[i], cand[i]) : 0ull;
        if (i < C) sk[i] = mine[e];
    }
    if (tid == 0) ek_sh = 0.0f;
    __syncthreads();
    const int live_e = C > 3072u ? 4 : (C > 2048u ? 3 : (C > 1024u ? 2 : 1));
    for (uint32_t j = 0; j < C; ++j) {
        const unsigned long long o = sk[j];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < live_e) rank[e] += o > mine[e] ? 1u : 0u;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint32_t i = (uint32_t)tid + 1024u * e;
        if (i < C && rank[e] < n_out) {
            float sc = key_score((uint32_t)(mine[e] >> 32));
            if (rank[e] == n_out - 1u) ek_sh = sc;
            if (METRIC == 1 && sc == 0.0f) sc = -0.0f;  // -(sqrt(0)) in the reference
            out_rows[rank[e]] = 0xFFFFFFFFu - (uint32_t)(mine[e] & 0xFFFFFFFFull);
            out_scores[rank[e]] = sc;
        }
    }
    __syncthreads();
    if (tid == 0) {
        if (status == ST_OK && C < live && n_out > 0)
            status = certify<METRIC>(ctl, d, ctl->na, ctl->t_key, ek_sh, 4);
        *out_n = status == ST_OK ? n_out : 0u;
        ctl->status = status;
        if (out_status) *out_status = status;
        // the fields a query accumulates into (the histograms were cleared by the stage before)
        ctl->resolved = 0;
        ctl->cand_count = 0;
        ctl->cut_ties = 0;
        ctl->tie_quota = 0;
        ctl->b1 = ctl->above1 = ctl->b2 = ctl->above2 = ctl->above3 = 0;
    }
}

// Merge of per-shard top-k lists (global row ids) that an all-gather left in
// device memory: one workgroup sorts all entries by (score key, row).
__global__ void __launch_bounds__(1024)
merge_topk_kernel(const uint32_t* __restrict__ rows, const float* __restrict__ scores,
                  const uint32_t* __restrict__ list_n, uint32_t n_lists, uint32_t k_in,
                  uint32_t k_out, uint32_t* __restrict__ out_rows, float* __restrict__ out_scores,
                  uint32_t* __restrict__ out_n) {
    __shared__ unsigned long long sk[CAND_CAP];
    __shared__ uint32_t total_sh;
    const int tid = threadIdx.x;
    const uint32_t C = n_lists * k_in;
    uint32_t P = 64;
    while (P < C) P <<= 1;
    if (tid == 0) total_sh = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t i = tid; i < P; i += 1024u) {
        unsigned long long v = 0ull;
        if (i < C) {
            const uint32_t l = i / k_in, j = i - l * k_in;
            const uint32_t n = list_n ? min(list_n[l], k_in) : k_in;
            if (j < n) {
                v = rank_key(score_key(scores[i]), rows[i]);
                mine += 1;
            }
        }
        sk[i] = v;
    }
    if (mine) atomicAdd(&total_sh, mine);
    for (uint32_t size = 2; size <= P; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (uint32_t i = tid; i < P / 2; i += 1024u) {
                const uint32_t pos = 2 * i - (i & (stride - 1));
                const unsigned long long a = sk[pos], b = sk[pos + stride];
                const bool desc = (pos & size) == 0;
                if (desc ? (a < b) : (a > b)) { sk[pos] = b; sk[pos + stride] = a; }
            }
        }
    }
    __syncthreads();
    const uint32_t total = total_sh;
    const uint32_t n_out = k_out < total ? k_out : total;
    for (uint32_t i = tid; i < n_out; i += 1024u) {
        const unsigned long long v = sk[i];
        out_rows[i] = 0xFFFFFFFFu - (uint32_t)(v & 0xFFFFFFFFull);
        out_scores[i] = key_score((uint32_t)(v >> 32));
    }
    if (tid == 0) *out_n = n_out;
}

// Exchange record of one query: [k score bits | k global rows | count]; the count word carries
// the query's status on this shard in bits 16..23 (0 = final). Global row = row0 + local * row_mul
// (contiguous shards: row_mul 1; shards striped over G ranks: row0 = rank, row_mul = G).
__global__ void __launch_bounds__(256)
pack_topk_kernel(const uint32_t* __restrict__ rows, const float* __restrict__ scores,
                 const uint32_t* __restrict__ n, const uint32_t* __restrict__ status, uint32_t row0,
                 uint32_t row_mul, uint32_t n_queries, uint32_t k, uint32_t* __restrict__ packed) {
    const uint32_t rec = 2u * k + 1u;
    const uint32_t total = n_queries * rec;
    for (uint32_t e = blockIdx.x * 256u + threadIdx.x; e < total; e += gridDim.x * 256u) {
        const uint32_t q = e / rec, j = e - q * rec;
        uint32_t v;
        if (j < k) v = __float_as_uint(scores[(size_t)q * k + j]);
        else if (j < 2u * k) v = rows[(size_t)q * k + (j - k)] * row_mul + row0;
        else v = (n[q] & 0xFFFFu) | (status ? (status[q] & 0xFFu) << 16 : 0u);
        packed[e] = v;
    }
}

// One workgroup per query: sort the gathered records of all shards.
__global__ void __launch_bounds__(1024)
merge_packed_kernel(const uint32_t* __restrict__ gathered, uint32_t n_lists, uint32_t n_queries,
                    uint32_t k, uint32_t* __restrict__ out_rows, float* __restrict__ out_scores,
                    uint32_t* __restrict__ out_n, uint32_t* __restrict__ out_status) {
    __shared__ unsigned long long sk[CAND_CAP];
    __shared__ uint32_t total_sh;
    const int tid = threadIdx.x;
    const uint32_t q = blockIdx.x;
    const uint32_t rec = 2u * k + 1u;
    const uint32_t C = n_lists * k;
    uint32_t P = 64;
    while (P < C) P <<= 1;
    if (tid == 0) total_sh = 0;
    __syncthreads();
    if (tid == 0 && out_status) {   // a query is final only if it is final on every shard
        uint32_t worst = 0;
        for (uint32_t l = 0; l < n_lists; ++l)
            worst = max(worst, (gathered[((size_t)l * n_queries + q) * rec + 2u * k] >> 16) & 0xFFu);
        out_status[q] = worst;
    }
    uint32_t mine = 0;
    for (uint32_t i = tid; i < P; i += 1024u) {
        unsigned long long v = 0ull;
        if (i < C) {
            const uint32_t l = i / k, j = i - l * k;
            const uint32_t* r = gathered + ((size_t)l * n_queries + q) * rec;
            const uint32_t cnt = min(r[2u * k] & 0xFFFFu, k);
            if (j < cnt) {
                v = rank_key(score_key(__uint_as_float(r[j])), r[k + j]);
                mine += 1;
            }
        }
        sk[i] = v;
    }
    if (mine) atomicAdd(&total_sh, mine);
    for (uint32_t size = 2; size <= P; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (uint32_t i = tid; i < P / 2; i += 1024u) {
                const uint32_t pos = 2 * i - (i & (stride - 1));
                const unsigned long long a = sk[pos], b = sk[pos + stride];
                const bool desc = (pos & size) == 0;
                if (desc ? (a < b) : (a > b)) { sk[pos] = b; sk[pos + stride] = a; }
            }
        }
    }
    __syncthreads();
    const uint32_t total = total_sh;
    const uint32_t n_out = k < total ? k : total;
    for (uint32_t i = tid; i < n_out; i += 1024u) {
        const unsigned long long v = sk[i];
        out_rows[(size_t)q * k + i] = 0xFFFFFFFFu - (uint32_t)(v & 0xFFFFFFFFull);
        out_scores[(size_t)q * k + i] = key_score((uint32_t)(v >> 32));
    }
    if (tid == 0) out_n[q] = n_out;
}

// ---------------------------------------------------------------------------
// Synthetic corpus: counter-based components, then a sequential-sum L2
// normalisation per row (two passes so that the stores are coalesced).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
synth_norm_kernel(unsigned long long seed, unsigned long long row0, unsigned long long row_step,
                  unsigned long long n, uint32_t d, float* __restrict__ norms) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x;
    if (i >= n) return;
    float ss = 0.0f;
    for (uint32_t c = 0; c < d; ++c) {
        const float v = synth_component(seed, row0 + i * row_step, c);
        const float p = v * v;
        ss = ss + p;
    }
    norms[i] = sqrtf(ss);
}

__global__ void __launch_bounds__(256)
synth_fill_kernel(unsigned long long seed, unsigned long long row0, unsigned long long row_step,
                  unsigned long long n, uint32_t d, const float* __restrict__ norms, float* __restrict__ rows) {
    const unsigned long long total = n * (unsigned long long)d;
    for (unsigned long long e = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; e < total;
         e += (unsigned long long)gridDim.x * 256ull) {
        const unsigned long long i = e / d;
        const uint32_t c = (uint32_t)(e - i * d);
        float v = synth_component(seed, row0 + i * row_step, c);
        const float nrm = norms[i];
        if (nrm > 0.0f) v = v / nrm;
        rows[e] = v;
    }
}

// Upper bound of max over rows of sum v*v (any summation order; inflated by
// 1e-3) into vmax_sq[0], and of max over rows of ||bf16(v) - v||^2 / ||v||^2 (inflated by
// 1 %; rows whose norm is outside the bf16 bound's range are left out: they are candidates
// whatever their approximate score) into vmax_sq[1]. One wave per row, coalesced.
__global__ void __launch_bounds__(256)
vmax_kernel(const float* __restrict__ rows, unsigned long long n, uint32_t d,
            float* __restrict__ vmax_sq, float* __restrict__ norm2_out) {
    const int lane = threadIdx.x & 63;
    const unsigned long long w0 = ((unsigned long long)blockIdx.x * 256ull + threadIdx.x) >> 6;
    const unsigned long long nw = ((unsigned long long)gridDim.x * 256ull) >> 6;
    float best = 0.0f, best_rho = 0.0f;
    for (unsigned long long r = w0; r < n; r += nw) {
        const float* v = rows + r * d;
        float s = 0.0f, e2 = 0.0f;
        for (uint32_t c = lane; c < d; c += 64) {
            const float x = v[c];
            s = __builtin_fmaf(x, x, s);
            e2 = bf16_round_err_sq(x, e2);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s += __shfl_xor(s, o);
            e2 += __shfl_xor(e2, o);
        }
        if (lane == 0 && norm2_out) norm2_out[r] = s;  // squared norm, reassociated sum
        if (s >= 0x1p-100f && s < INFINITY) best_rho = fmaxf(best_rho, e2 / s * 1.01f);  // (fmaxf drops a NaN)
        s = s * 1.001f;
        if (!(s < INFINITY)) continue;  // non-finite rows are always re-scored exactly
        best = fmaxf(best, s);
    }
    if (lane == 0 && best > 0.0f)
        atomicMax(reinterpret_cast<unsigned int*>(vmax_sq), __float_as_uint(best));
    if (lane == 0 && best_rho > 0.0f)
        atomicMax(reinterpret_cast<unsigned int*>(vmax_sq) + 1, __float_as_uint(best_rho));
}

// Reference point for the roofline figures: a plain contiguous stream over the same memory,
// 16-byte non-temporal loads, eight per thread in flight, nothing but an add per load.
__global__ void __launch_bounds__(256)
stream_probe_kernel(const v4f* __restrict__ p, unsigned long long n4, float* __restrict__ sink) {
    v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
    const unsigned long long stride = (unsigned long long)gridDim.x * 256ull * 8ull;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256ull * 8ull + threadIdx.x; i < n4; i += stride) {
        v4f x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned long long j = i + (unsigned long long)u * 256ull;
            x[u] = j < n4 ? __builtin_nontemporal_load(p + j) : acc;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += x[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;  // keeps the loads alive
}

// Zeroes two word ranges in one launch (the no-allocation entry's accumulators and published wave
// bests: a kernel, not hipMemsetAsync -- captured into a HIP graph, a memset node in front of the
// scan was seen to run unordered with it on every replay after the first).
__global__ void __launch_bounds__(256) zero_words_kernel(uint32_t* __restrict__ a, uint32_t na, uint32_t* __restrict__ b, uint32_t nb) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < na) a[i] = 0u;
    else if (i - na < nb) b[i - na] = 0u;
}

hipError_t launch_zero_words(uint32_t* a, uint32_t na, uint32_t* b, uint32_t nb, hipStream_t st) {
    const uint32_t n = na + nb;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(zero_words_kernel, dim3((n + 255u) / 256u), dim3(256), 0, st, a, na, b, nb);
    return hipGetLastError();
}

hipError_t launch_stream_probe(const float* d_rows, uint64_t n_floats, float* d_sink, hipStream_t st) {
    hipLaunchKernelGGL(stream_probe_kernel, dim3(2048), dim3(256), 0, st, reinterpret_cast<const v4f*>(d_rows),
                       (unsigned long long)(n_floats / 4), d_sink);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
static uint32_t grid_for(uint64_t items, uint32_t per_block, uint32_t cap) {
    uint64_t g = (items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (uint32_t)g;
}

template <int METRIC>
static uint32_t scan_exact_dispatch(const float* d_q, const ScanArgs& a, const Scratch& s, uint32_t need,
                                    uint32_t k, const float* d_vmax_sq, hipStream_t st, const PrevSelect* prev) {
    if ((a.dim & 3u) == 0u) {
        const uint32_t n_tiles = (a.n_rows + 63u) / 64u;
        // One workgroup of 4 waves per CU, each wave with one 8 KB chunk in flight,
        // measured best on MI355X (10M x 384: 6.66 TB/s; 2 or 3 workgroups per CU 6.51 / 6.43,
        // one workgroup of 3 / 5 / 6 / 8 waves 6.35 / 6.51 / 6.50 / 6.47,
        // 64-float chunks 6.44 (with 2 / 3 waves per CU 5.93 / 6.26), default-policy instead of
        // non-temporal loads 5.94; buffer loads with any of the sc0 / sc1 / nt policies at best 6.25 (nt)).
        // Short scans agree: 1 M x 384 with 1 / 2 / 3 workgroups per CU 0.2375 / 0.243 / 0.259 ms, 1.25 M
        // 0.295 / 0.3015 / 0.3256. Very short ones (10 k - 500 k rows) are their bytes at the streaming
        // rate plus some 12 us that do not shrink: a variant with four chunks of loads in flight per wave
        // and the query in LDS instead of the scalar cache, with up to three workgroups per CU, moved
        // 10 k rows from 16.8 to 15.1 us, 100 k from 34.7 to 32.6 (22 us of it the bytes), 250 k and 500 k
        // not at all, and was not kept.
        static int n_cu = 0;
        if (n_cu == 0) {
            int dev = 0;
            n_cu = 256;
            if (hipGetDevice(&dev) == hipSuccess)
                (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
        }
        uint32_t grid = grid_for(n_tiles, 4, (uint32_t)n_cu);
        // Every wave walks its tiles in rounds (tile t belongs to wave t % n_waves), and the last
        // round costs a whole round however few waves still have a tile in it: on a short scan
        // (1 M rows = 15.26 rounds of 1024 waves) that is 5 % of the time for a quarter-full round.
        // The pass is bound by memory, not by the number of workgroups in flight, so a few
        // workgroups fewer that fill their last round do better: among 15/16 .. 16/16 of the CUs
        // take the count that leaves the fewest empty tile slots.
        if (grid == (uint32_t)n_cu && !getenv("TRAG_SCAN_FULL_GRID")) {
            uint64_t best_slots = ~0ull;
            uint32_t best = grid;
            for (uint32_t g = (uint32_t)n_cu; g >= (uint32_t)n_cu - (uint32_t)n_cu / 16u && g > 0; --g) {
                const uint64_t waves = 4ull * g, slots = (n_tiles + waves - 1) / waves * waves;
                if (slots < best_slots) { best_slots = slots; best = g; }
            }
            grid = best;
        }
        const uint32_t ep = (s.wbest && sparse_scan_pays(n_tiles, grid * 4u)) ? s.epoch : 0u;
        // load-by-load refill pays only for short rows (same run, MI355X: d = 128 +6 %, 384 -2 %,
        // 768 / 1536 / 4096 within noise)
        const PrevSelect none{};
        if (prev) {   // one more workgroup, in front: the previous query's select
            if (a.dim <= 256)
                hipLaunchKernelGGL((scan_exact_v4_kernel<METRIC, 32, true, 4, true, true>), dim3(grid + 1u), dim3(256), 0,
                                   st, a.corpus, d_q, a.dead, a.n_rows, a.dim, need, k, d_vmax_sq, s.ctl, s.keys,
                                   s.tmax, s.wmax, s.hist, s.wbest, ep, *prev);
            else
                hipLaunchKernelGGL((scan_exact_v4_kernel<METRIC, 32, true, 4, false, true>), dim3(grid + 1u), dim3(256), 0,
                                   st, a.corpus, d_q, a.dead, a.n_rows, a.dim, need, k, d_vmax_sq, s.ctl, s.keys,
                                   s.tmax, s.wmax, s.hist, s.wbest, ep, *prev);
        } else if (a.dim <= 256)
            hipLaunchKernelGGL((scan_exact_v4_kernel<METRIC, 32, true, 4, true>), dim3(grid), dim3(256), 0, st,
                               a.corpus, d_q, a.dead, a.n_rows, a.dim, need, k, d_vmax_sq, s.ctl, s.keys,
                               s.tmax, s.wmax, s.hist, s.wbest, ep, none);
        else
            hipLaunchKernelGGL((scan_exact_v4_kernel<METRIC, 32, true, 4, false>), dim3(grid), dim3(256), 0, st,
                               a.corpus, d_q, a.dead, a.n_rows, a.dim, need, k, d_vmax_sq, s.ctl, s.keys,
                               s.tmax, s.wmax, s.hist, s.wbest, ep, none);
        return grid * 4u;
    } else {
        const uint32_t grid = grid_for(a.n_rows, 256, 256 * 8);
        hipLaunchKernelGGL(scan_exact_any_kernel<METRIC>, dim3(grid), dim3(256), 0, st, a.corpus,
                           d_q, a.dead, a.n_rows, a.dim, need, k, d_vmax_sq, s.ctl, s.keys, s.tmax, s.wmax, s.hist);
        return grid * 4u;
    }
}

// prev (optional; rows of 4 k components only -- scan_can_fuse): the select of the handle's previous
// query rides in this launch (see scan_exact_v4_kernel, FUSE).
bool scan_can_fuse(uint32_t dim) { return (dim & 3u) == 0u; }

hipError_t launch_scan_exact(const float* d_q, const ScanArgs& a, const Scratch& s, uint32_t need,
                             uint32_t k, const float* d_vmax_sq, uint32_t* n_waves, hipStream_t st,
                             const PrevSelect* prev) {
    switch (a.metric) {
    case TRAG_COSINE: *n_waves = scan_exact_dispatch<0>(d_q, a, s, need, k, d_vmax_sq, st, prev); break;
    case TRAG_EUCLIDEAN: *n_waves = scan_exact_dispatch<1>(d_q, a, s, need, k, d_vmax_sq, st, prev); break;
    default: *n_waves = scan_exact_dispatch<2>(d_q, a, s, need, k, d_vmax_sq, st, prev); break;
    }
    return hipGetLastError();
}

hipError_t launch_select(const ScanArgs& a, const Scratch& s, hipStream_t st) {
    // The refinement passes are needed only when one coarse bin holds more rows
    // than the final sort can take; otherwise they return at once, so keep
    // their grids small.
    const uint32_t rgrid = grid_for(a.n_rows, 256 * 16, 256);
    hipLaunchKernelGGL(refine_kernel<2>, dim3(rgrid), dim3(256), 0, st, s.keys, a.n_rows, s.ctl,
                       s.hist);
    hipLaunchKernelGGL(refine_kernel<3>, dim3(rgrid), dim3(256), 0, st, s.keys, a.n_rows, s.ctl,
                       s.hist);
    const uint32_t n_tiles = (a.n_rows + 63u) / 64u;
    const uint32_t cgrid = grid_for(n_tiles, 256, 256);
    hipLaunchKernelGGL(compact_kernel, dim3(cgrid), dim3(256), 0, st, s.keys, s.tmax, a.n_rows,
                       s.ctl, s.hist, s.cand, (size_t)0, (size_t)0);
    return hipGetLastError();
}

template <int METRIC>
static void final_dispatch(const float* d_q, const ScanArgs& a, const Scratch& s, bool rescore,
                           uint32_t live, uint32_t* d_out_rows, float* d_out_scores,
                           uint32_t* d_out_n, uint32_t* d_status, hipStream_t st) {
    if (rescore)
        hipLaunchKernelGGL((final_kernel<METRIC, true>), dim3(1), dim3(1024), 0, st, a.corpus, d_q,
                           a.n_rows, a.dim, live, s.ctl, s.keys, s.cand, d_out_rows, d_out_scores,
                           d_out_n, d_status, (size_t)0, 0u, 0);
    else
        hipLaunchKernelGGL((final_kernel<METRIC, false>), dim3(1), dim3(1024), 0, st, a.corpus, d_q,
                           a.n_rows, a.dim, live, s.ctl, s.keys, s.cand, d_out_rows, d_out_scores,
                           d_out_n, d_status, (size_t)0, 0u, 0);
}

hipError_t launch_final(const float* d_q, const ScanArgs& a, const Scratch& s, bool rescore,
                             uint32_t live, uint32_t* d_out_rows, float* d_out_scores,
                             uint32_t* d_out_n, uint32_t* d_status, hipStream_t st) {
    switch (a.metric) {
    case TRAG_COSINE:
        final_dispatch<0>(d_q, a, s, rescore, live, d_out_rows, d_out_scores, d_out_n, d_status, st);
        break;
    case TRAG_EUCLIDEAN:
        final_dispatch<1>(d_q, a, s, rescore, live, d_out_rows, d_out_scores, d_out_n, d_status, st);
        break;
    default:
        final_dispatch<2>(d_q, a, s, rescore, live, d_out_rows, d_out_scores, d_out_n, d_status, st);
        break;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Large k (more results than the one-workgroup sort holds): sort every row.
// rank keys of all rows -> global bitonic network, one launch per step.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
rank_keys_kernel(const uint32_t* __restrict__ keys, uint32_t n_rows, unsigned long long padded,
                 unsigned long long* __restrict__ out) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; i < padded;
         i += (unsigned long long)gridDim.x * 256ull) {
        unsigned long long v = 0ull;
        if (i < n_rows) {
            const uint32_t key = keys[i];
            if (key) v = rank_key(key, (uint32_t)i);  // removed rows sort last as 0
        }
        out[i] = v;
    }
}

__global__ void __launch_bounds__(256)
bitonic_step_kernel(unsigned long long* __restrict__ a, unsigned long long padded,
                    unsigned long long size, unsigned long long stride) {
    const unsigned long long half = padded >> 1;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; i < half;
         i += (unsigned long long)gridDim.x * 256ull) {
        const unsigned long long pos = 2ull * i - (i & (stride - 1ull));
        const unsigned long long x = a[pos], y = a[pos + stride];
        const bool desc = (pos & size) == 0ull;
        if (desc ? (x < y) : (x > y)) { a[pos] = y; a[pos + stride] = x; }
    }
}

template <int METRIC>
__global__ void __launch_bounds__(256)
emit_sorted_kernel(const unsigned long long* __restrict__ sorted, uint32_t n_out,
                   uint32_t* __restrict__ out_rows, float* __restrict__ out_scores,
                   uint32_t* __restrict__ out_n, uint32_t* __restrict__ out_status) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n_out; i += gridDim.x * 256u) {
        const unsigned long long v = sorted[i];
        out_rows[i] = 0xFFFFFFFFu - (uint32_t)(v & 0xFFFFFFFFull);
        float sc = key_score((uint32_t)(v >> 32));
        if (METRIC == 1 && sc == 0.0f) sc = -0.0f;
        out_scores[i] = sc;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *out_n = n_out;
        if (out_status) *out_status = ST_OK;
    }
}

hipError_t launch_full_sort(const ScanArgs& a, const Scratch& s, unsigned long long* d_sort,
                            unsigned long long padded, uint32_t n_out, uint32_t* d_out_rows,
                            float* d_out_scores, uint32_t* d_out_n, uint32_t* d_status,
                            hipStream_t st) {
    const uint32_t grid = grid_for(padded, 256 * 4, 256 * 8);
    hipLaunchKernelGGL(rank_keys_kernel, dim3(grid), dim3(256), 0, st, s.keys, a.n_rows, padded, d_sort);
    for (unsigned long long size = 2; size <= padded; size <<= 1)
        for (unsigned long long stride = size >> 1; stride > 0; stride >>= 1)
            hipLaunchKernelGGL(bitonic_step_kernel, dim3(grid), dim3(256), 0, st, d_sort, padded, size,
                               stride);
    const uint32_t egrid = grid_for(n_out, 256, 1024);
    if (a.metric == TRAG_EUCLIDEAN)
        hipLaunchKernelGGL(emit_sorted_kernel<1>, dim3(egrid), dim3(256), 0, st, d_sort, n_out,
                           d_out_rows, d_out_scores, d_out_n, d_status);
    else
        hipLaunchKernelGGL(emit_sorted_kernel<0>, dim3(egrid), dim3(256), 0, st, d_sort, n_out,
                           d_out_rows, d_out_scores, d_out_n, d_status);
    return hipGetLastError();
}

// After the multi-kernel select: clear the fields of the control block that a
// query accumulates into (the histograms are cleared with a memset).
__global__ void reset_ctl_kernel(SearchCtl* ctl) {
    ctl->resolved = 0;
    ctl->cand_count = 0;
    ctl->cut_ties = 0;
    ctl->tie_quota = 0;
    ctl->b1 = ctl->above1 = ctl->b2 = ctl->above2 = ctl->above3 = 0;
}

hipError_t launch_reset_ctl(const Scratch& s, hipStream_t st) {
    hipLaunchKernelGGL(reset_ctl_kernel, dim3(1), dim3(1), 0, st, s.ctl);
    return hipGetLastError();
}

template <int METRIC>
static void select_final_dispatch(const float* d_q, const ScanArgs& a, const Scratch& s, bool rescore,
                                  uint32_t live, uint32_t need, uint32_t k, uint32_t n_waves,
                                  uint32_t* d_out_rows, float* d_out_scores, uint32_t* d_out_n,
                                  uint32_t* d_status, uint32_t n_queries, const MultiStride& ms,
                                  uint32_t out_stride, hipStream_t st) {
    if (rescore)
        hipLaunchKernelGGL((select_final_kernel<METRIC, true>), dim3(n_queries), dim3(1024), 0, st, a.corpus,
                           d_q, a.n_rows, a.dim, live, need, k, n_waves, s.ctl, s.hist, s.keys, s.tmax,
                           s.wmax, d_out_rows, d_out_scores, d_out_n, d_status, ms, out_stride, s.events,
                           (const unsigned long long*)nullptr, 0u);
    else
        hipLaunchKernelGGL((select_final_kernel<METRIC, false>), dim3(n_queries), dim3(1024), 0, st, a.corpus,
                           d_q, a.n_rows, a.dim, live, need, k, n_waves, s.ctl, s.hist, s.keys, s.tmax,
                           s.wmax, d_out_rows, d_out_scores, d_out_n, d_status, ms, out_stride, s.events,
                           n_queries == 1 ? s.wbest : nullptr, (n_queries == 1 && s.wbest) ? s.epoch : 0u);
}

hipError_t launch_select_final(const float* d_q, const ScanArgs& a, const Scratch& s, bool rescore,
                               uint32_t live, uint32_t need, uint32_t k, uint32_t n_waves,
                               uint32_t* d_out_rows, float* d_out_scores, uint32_t* d_out_n,
                               uint32_t* d_status, hipStream_t st, uint32_t n_queries,
                               const MultiStride* ms, uint32_t out_stride) {
    const MultiStride m = ms ? *ms : MultiStride{};
    switch (a.metric) {
    case TRAG_COSINE:
        select_final_dispatch<0>(d_q, a, s, rescore, live, need, k, n_waves, d_out_rows, d_out_scores, d_out_n, d_status, n_queries, m, out_stride, st);
        break;
    case TRAG_EUCLIDEAN:
        select_final_dispatch<1>(d_q, a, s, rescore, live, need, k, n_waves, d_out_rows, d_out_scores, d_out_n, d_status, n_queries, m, out_stride, st);
        break;
    default:
        select_final_dispatch<2>(d_q, a, s, rescore, live, need, k, n_waves, d_out_rows, d_out_scores, d_out_n, d_status, n_queries, m, out_stride, st);
        break;
    }
    return hipGetLastError();
}

// Stages after a mirror scan. One-workgroup form: prune -> exact scores -> order.
// Multi-level form (crowded score bins): radix levels -> threshold -> compact -> exact
// scores -> order; the caller clears the histograms afterwards.
hipError_t launch_select_mirror(const float* d_q, const ScanArgs& a, const Scratch& s, bool multi_level,
                                uint32_t live, uint32_t need, uint32_t k, uint32_t n_waves,
                                uint32_t* d_out_rows, float* d_out_scores, uint32_t* d_out_n,
                                uint32_t* d_status, hipStream_t st) {
    const uint32_t rgrid = grid_for(a.n_rows, 256 * 16, 256);
    const uint32_t n_tiles = (a.n_rows + 63u) / 64u;
    const uint32_t cgrid = grid_for(n_tiles, 256, 256);
    if (multi_level) {
        hipLaunchKernelGGL(refine_kernel<2>, dim3(rgrid), dim3(256), 0, st, s.keys, a.n_rows, s.ctl, s.hist);
        hipLaunchKernelGGL(refine_kernel<3>, dim3(rgrid), dim3(256), 0, st, s.keys, a.n_rows, s.ctl, s.hist);
    }
#define TRAG_MIRROR_STAGES(METRIC)                                                                         \
    do {                                                                                                   \
        if (multi_level) {                                                                                 \
            hipLaunchKernelGGL(mirror_threshold_kernel<METRIC>, dim3(1), dim3(256), 0, st, s.ctl, s.hist,  \
                               a.dim);                                                                     \
            hipLaunchKernelGGL(compact_kernel, dim3(cgrid), dim3(256), 0, st, s.keys, s.tmax, a.n_rows,    \
                               s.ctl, (const uint32_t*)nullptr, s.cand, (size_t)0, (size_t)0);             \
        } else {                                                                                           \
            hipLaunchKernelGGL(select_mirror_kernel<METRIC>, dim3(1), dim3(1024), 0, st, a.n_rows, a.dim,  \
                               live, need, k, n_waves, s.ctl, s.hist, s.keys, s.tmax, s.wmax, s.cand,      \
                               s.events, s.wbest, s.wbest ? s.epoch : 0u);                                 \
        }                                                                                                  \
        hipLaunchKernelGGL(rescore_wave_kernel<METRIC>, dim3(256), dim3(256), 0, st, a.corpus, d_q,        \
                           a.n_rows, a.dim, s.ctl, s.cand, s.ckeys);                                       \
        hipLaunchKernelGGL(mirror_final_kernel<METRIC>, dim3(1), dim3(1024), 0, st, a.dim, live, s.ctl,    \
                           s.cand, s.ckeys, d_out_rows, d_out_scores, d_out_n, d_status);                  \
    } while (0)
    if (a.metric == TRAG_COSINE) TRAG_MIRROR_STAGES(0);
    else if (a.metric == TRAG_EUCLIDEAN) TRAG_MIRROR_STAGES(1);
    else TRAG_MIRROR_STAGES(2);
#undef TRAG_MIRROR_STAGES
    return hipGetLastError();
}

// Queries one multi-query pass can take at this dimension (their LDS copy must fit), 0 or 1 = none.
uint32_t multi_group_limit(uint32_t dim) {
    if (dim & 3u) return 0;
    uint32_t nq = MULTI_QUERIES;
    while (nq >= 2 && multi_static_lds(nq) + (uint64_t)nq * dim * sizeof(float) > MULTI_LDS_TOTAL) --nq;
    return nq;
}

template <int METRIC>
static uint32_t scan_multi_dispatch(const float* d_q, uint32_t nq, const ScanArgs& a, const Scratch& s,
                                    const MultiStride& ms, uint32_t need, uint32_t k,
                                    const float* d_vmax_sq, hipStream_t st) {
    const uint32_t n_tiles = (a.n_rows + 63u) / 64u;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        n_cu = 256;
        if (hipGetDevice(&dev) == hipSuccess)
            (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    }
    const uint32_t grid = grid_for(n_tiles, 4, (uint32_t)n_cu);
    const size_t q_bytes = (size_t)nq * a.dim * sizeof(float);
#define TRAG_LAUNCH_MULTI(NQ)                                                                              \
    do {                                                                                                   \
        static bool attr_set = false;                                                                      \
        if (!attr_set) {                                                                                   \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_exact_multi_kernel<METRIC, NQ>), \
                                      hipFuncAttributeMaxDynamicSharedMemorySize,                          \
                                      (int)(MULTI_LDS_TOTAL - multi_static_lds(NQ)));                      \
            attr_set = true;                                                                               \
        }                                                                                                  \
        hipLaunchKernelGGL((scan_exact_multi_kernel<METRIC, NQ>), dim3(grid), dim3(256), q_bytes, st,      \
                           a.corpus, d_q, a.dead, a.n_rows, a.dim, need, k, d_vmax_sq, s.ctl, s.keys,      \
                           s.tmax, s.wmax, s.hist, ms);                                                    \
    } while (0)
    switch (nq) {
    case 2: TRAG_LAUNCH_MULTI(2); break;
    case 3: TRAG_LAUNCH_MULTI(3); break;
    case 4: TRAG_LAUNCH_MULTI(4); break;
    case 5: TRAG_LAUNCH_MULTI(5); break;
    case 6: TRAG_LAUNCH_MULTI(6); break;
    case 7: TRAG_LAUNCH_MULTI(7); break;
    default: TRAG_LAUNCH_MULTI(8); break;
    }
#undef TRAG_LAUNCH_MULTI
    return grid * 4u;
}

hipError_t launch_scan_exact_multi(const float* d_q, uint32_t n_queries, const ScanArgs& a,
                                   const Scratch& s, const MultiStride& ms, uint32_t need, uint32_t k,
                                   const float* d_vmax_sq, uint32_t* n_waves, hipStream_t st) {
    if (n_queries < 2 || n_queries > multi_group_limit(a.dim)) return hipErrorInvalidValue;
    switch (a.metric) {
    case TRAG_COSINE: *n_waves = scan_multi_dispatch<0>(d_q, n_queries, a, s, ms, need, k, d_vmax_sq, st); break;
    case TRAG_EUCLIDEAN: *n_waves = scan_multi_dispatch<1>(d_q, n_queries, a, s, ms, need, k, d_vmax_sq, st); break;
    default: *n_waves = scan_multi_dispatch<2>(d_q, n_queries, a, s, ms, need, k, d_vmax_sq, st); break;
    }
    return hipGetLastError();
}

hipError_t launch_fill_synthetic(float* d_rows, uint64_t row0, uint64_t n, uint32_t dim,
                                 uint64_t seed, float* d_norm_tmp, hipStream_t st, uint64_t row_step) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(synth_norm_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, seed,
                       row0, row_step, n, dim, d_norm_tmp);
    const uint32_t grid = grid_for(n * dim, 256 * 8, 256 * 16);
    hipLaunchKernelGGL(synth_fill_kernel, dim3(grid), dim3(256), 0, st, seed, row0, row_step, n, dim,
                       d_norm_tmp, d_rows);
    return hipGetLastError();
}

hipError_t launch_merge_topk(const uint32_t* d_rows, const float* d_scores, const uint32_t* d_list_n,
                             uint32_t n_lists, uint32_t k_in, uint32_t k_out, uint32_t* d_out_rows,
                             float* d_out_scores, uint32_t* d_out_n, hipStream_t st) {
    hipLaunchKernelGGL(merge_topk_kernel, dim3(1), dim3(1024), 0, st, d_rows, d_scores, d_list_n,
                       n_lists, k_in, k_out, d_out_rows, d_out_scores, d_out_n);
    return hipGetLastError();
}

hipError_t launch_pack_topk(const uint32_t* d_rows, const float* d_scores, const uint32_t* d_n,
                            uint32_t row0, uint32_t n_queries, uint32_t k, uint32_t* d_packed,
                            hipStream_t st, uint32_t row_mul, const uint32_t* d_status) {
    const uint32_t grid = grid_for((uint64_t)n_queries * (2 * k + 1), 256, 1024);
    hipLaunchKernelGGL(pack_topk_kernel, dim3(grid), dim3(256), 0, st, d_rows, d_scores, d_n, d_status, row0,
                       row_mul, n_queries, k, d_packed);
    return hipGetLastError();
}

hipError_t launch_merge_packed(const uint32_t* d_gathered, uint32_t n_lists, uint32_t n_queries,
                               uint32_t k, uint32_t* d_out_rows, float* d_out_scores,
                               uint32_t* d_out_n, hipStream_t st, uint32_t* d_out_status) {
    hipLaunchKernelGGL(merge_packed_kernel, dim3(n_queries), dim3(1024), 0, st, d_gathered, n_lists,
                       n_queries, k, d_out_rows, d_out_scores, d_out_n, d_out_status);
    return hipGetLastError();
}

hipError_t launch_update_vmax(const float* d_rows, uint64_t n, uint32_t dim, float* d_vmax_sq,
                              float* d_norm2_out, hipStream_t st) {
    if (n == 0) return hipSuccess;
    const uint32_t grid = grid_for(n, 4, 256 * 8);
    hipLaunchKernelGGL(vmax_kernel, dim3(grid), dim3(256), 0, st, d_rows, n, dim, d_vmax_sq,
                       d_norm2_out);
    return hipGetLastError();
}

// Batched compaction and final stage: blockIdx.y / blockIdx.x select the query.
hipError_t launch_batch_finish(const float* d_Q, uint32_t n_queries, const ScanArgs& a,
                               const BatchScratch& s, uint32_t live, uint32_t out_stride,
                               uint32_t* d_out_rows, float* d_out_scores, uint32_t* d_out_n,
                               uint32_t* d_status, hipStream_t st) {
    const uint32_t n_tiles = (a.n_rows + 63u) / 64u;
    const uint32_t cgrid = grid_for(n_tiles, 256 * 4, 64);
    hipLaunchKernelGGL(compact_kernel, dim3(cgrid, n_queries), dim3(256), 0, st, s.keys, s.tmax,
                       a.n_rows, s.ctls, (const uint32_t*)nullptr, s.cand, s.keys_stride,
                       s.tmax_stride);
    if (a.metric == TRAG_COSINE)
        hipLaunchKernelGGL((final_kernel<0, true>), dim3(n_queries), dim3(1024), 0, st, a.corpus, d_Q,
                           a.n_rows, a.dim, live, s.ctls, s.keys, s.cand, d_out_rows, d_out_scores,
                           d_out_n, d_status, s.keys_stride, out_stride, 1);
    else if (a.metric == TRAG_EUCLIDEAN)
        hipLaunchKernelGGL((final_kernel<1, true>), dim3(n_queries), dim3(1024), 0, st, a.corpus, d_Q,
                           a.n_rows, a.dim, live, s.ctls, s.keys, s.cand, d_out_rows, d_out_scores,
                           d_out_n, d_status, s.keys_stride, out_stride, 2);
    else
        hipLaunchKernelGGL((final_kernel<2, true>), dim3(n_queries), dim3(1024), 0, st, a.corpus, d_Q,
                           a.n_rows, a.dim, live, s.ctls, s.keys, s.cand, d_out_rows, d_out_scores,
                           d_out_n, d_status, s.keys_stride, out_stride, 1);
    return hipGetLastError();
}

hipError_t launch_bf16_finish(const float* d_Q, uint32_t n_queries, const ScanArgs& a, const Bf16Scratch& s,
                              uint32_t live, uint32_t out_stride, uint32_t* d_out_rows, float* d_out_scores,
                              uint32_t* d_out_n, uint32_t* d_status, hipStream_t st) {
    const bool gather = (a.dim & 3u) == 0u;  // the staged re-score needs 16-byte row pieces
#define TRAG_BF16_FINAL(METRIC)                                                                            \
    do {                                                                                                   \
        if (gather)                                                                                        \
            hipLaunchKernelGGL(rescore_kernel<METRIC>, dim3(n_queries, CAND_CAP / 256), dim3(256), 0, st,  \
                               a.corpus, d_Q, a.n_rows, a.dim, s.ctls, s.cand, s.ckeys);                   \
        hipLaunchKernelGGL((final_kernel<METRIC, true>), dim3(n_queries), dim3(1024), 0, st, a.corpus,     \
                           d_Q, a.n_rows, a.dim, live, s.ctls, (const uint32_t*)nullptr, s.cand,           \
                           d_out_rows, d_out_scores, d_out_n, d_status, (size_t)0, out_stride, 3,          \
                           gather ? (const uint32_t*)s.ckeys : (const uint32_t*)nullptr);                  \
    } while (0)
    if (a.metric == TRAG_COSINE) TRAG_BF16_FINAL(0);
    else if (a.metric == TRAG_EUCLIDEAN) TRAG_BF16_FINAL(1);
    else TRAG_BF16_FINAL(2);
#undef TRAG_BF16_FINAL
    return hipGetLastError();
}

}  // namespace trag
