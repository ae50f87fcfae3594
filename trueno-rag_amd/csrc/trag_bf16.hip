// Query batches with a bf16 matrix-core PREFILTER and an exact f32 finish.
//
// The returned rows and scores are the reference's f32 results, bit for bit:
// the bf16 pass only decides which rows can possibly be in a query's top k.
//
//   1. a sample of the corpus tiles (every tile_step-th) goes through the GEMM;
//      per query the k-th largest tile maximum A_k of the sample is a score that
//      at least k live rows reach approximately;
//   2. every approximate score is within eps of the reference's f32 score
//      (bf16 rounding of both operands, 2^-8 relative each, plus f32
//      accumulation; bound in trag_device.h), so the exact k-th best is >= A_k - eps and
//      every row of the exact top k has an approximate score >= A_k - 2 eps;
//   3. the full GEMM pass appends exactly those rows to the query's candidate
//      list (no score matrix is written: the pass reads the corpus once and
//      writes a few hundred row ids per query);
//   4. the candidates are re-scored with the reference arithmetic, sorted, and
//      the bound is checked once more against the exact k-th score (final_kernel).
// A candidate list that overflows, or a query outside the bound's range
// (non-finite norms), is repeated with the exact scan by the host.
//
// With 2.5 PFLOP/s of dense bf16 the pass is bound by the HBM read of the f32
// corpus (converted to bf16 on the way into LDS) up to several hundred queries.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "trag_device.h"
#include "trag_internal.h"

#ifndef BF_FRAG_SCHED
#define BF_FRAG_SCHED 2
#endif
#ifndef BF_FRAG_SCHED_R
#define BF_FRAG_SCHED_R 1
#endif
#ifndef BF_ASM_XFER
#define BF_ASM_XFER 1
#endif
#ifndef BF_PINGPONG
#define BF_PINGPONG 0
#endif
#ifndef BF16_THIN_SAMPLE
#define BF16_THIN_SAMPLE 1
#endif

namespace trag {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int BF_K = 64;   // f32 columns staged per step (= 4 MFMA k-steps of 16)
constexpr int BF_LD = 36;  // dwords per LDS row: 64 bf16 + 4 dwords of padding (conflict-free b128 reads)

constexpr int TAU_THREADS = 256;
static_assert(HIST_BINS == 2048u, "find_bin_256 assumes 2048 bins");

// Between the sample pass and the full pass, one workgroup per query.
// First the query's threshold in key space: the need-th largest of the sample's tile maxima (at
// least `need` sampled rows reach it), by three radix levels (11 + 11 + 10 bits) over the maxima,
// which a thread keeps in registers (up to 16 each; more are re-read per level). The maxima are keys
// of the row-dependent part v of the score (cosine v = dot * inv_nb = score * ||q||; dot product
// v = score; Euclidean v = 2 dot - ||v||^2 = -(squared distance) + ||q||^2).
// Then the threshold is lowered by 2 eps (see the header), published in v-space for the full pass
// and kept in score space (t_key) for the final check, and the ladder of the thresholds that tighten
// during the full pass (trag_device.h) is set up: level 0 is the threshold found here, a step is
// 2.02 eps / DYN_SUB in v-space; no ladder (step 0) for a query without a finite threshold. A
// query whose scale is outside the bound's range gets the threshold -inf: its list overflows and
// the host repeats it with the exact scan. Workgroups behind the batch write the padding of the
// ladder arrays (+inf: the passes read whole query groups).
// (256 threads, so that a CU holds eight queries at a time: the stage is three dependent rounds of
// barriers per query, not work. As two launches -- 1024 threads re-reading the maxima per level, then a
// kernel for the ladder -- it took 17 + 5.5 us for 1000 queries.)
template <int METRIC>
__global__ void __launch_bounds__(TAU_THREADS)
tau_shift_kernel(const uint32_t* __restrict__ tmax, size_t tmax_stride, uint32_t n_tiles,
                 SearchCtl* __restrict__ ctls, uint32_t n_queries, uint32_t padded, uint32_t d,
                 float* __restrict__ thr, uint32_t* __restrict__ dyn_lvl, float* __restrict__ dyn_lad,
                 uint32_t* __restrict__ dyn_cnt) {
    __shared__ uint32_t lhist[HIST_BINS];
    __shared__ uint32_t sh[6];
    const int tid = threadIdx.x;
    const uint32_t q = blockIdx.x;
    if (q >= n_queries) {
        const uint32_t i = n_queries + (q - n_queries) * (uint32_t)TAU_THREADS + (uint32_t)tid;
        if (i < padded) {
            dyn_lvl[i] = 0u;
            dyn_lad[2 * i] = INFINITY;
            dyn_lad[2 * i + 1] = 0.0f;
        }
        return;
    }
    const uint32_t* tm = tmax + (size_t)q * tmax_stride;
    SearchCtl* ctl = ctls + q;
    uint32_t need = ctl->need;
    for (uint32_t j = (uint32_t)tid; j < DYN_LEVELS; j += (uint32_t)TAU_THREADS) dyn_cnt[(size_t)q * DYN_LEVELS + j] = 0u;
    constexpr int KEEP = 16;
    uint32_t mine[KEEP];
#pragma unroll
    for (int j = 0; j < KEEP; ++j) {
        const uint32_t i = (uint32_t)j * TAU_THREADS + (uint32_t)tid;
        mine[j] = i < n_tiles ? tm[i] : 0u;
    }
    uint32_t prefix = 0, prefix_bits = 0;
    if (need <= n_tiles) {   // (fewer tiles than wanted rows: every live row, key 1)
        const int shifts[3] = {21, 10, 0};
        const uint32_t widths[3] = {11, 11, 10};
#pragma unroll
        for (int level = 0; level < 3; ++level) {
            for (int i = tid; i < (int)HIST_BINS; i += TAU_THREADS) lhist[i] = 0u;
            __syncthreads();
            const uint32_t mshift = 32u - prefix_bits;
            const uint32_t shift = (uint32_t)shifts[level], mask = (1u << widths[level]) - 1u;
#pragma unroll
            for (int j = 0; j < KEEP; ++j) {
                const uint32_t i = (uint32_t)j * TAU_THREADS + (uint32_t)tid;
                const bool match = prefix_bits == 0 || (mine[j] >> mshift) == prefix;
                if (i < n_tiles && match) atomicAdd(&lhist[(mine[j] >> shift) & mask], 1u);
            }
            for (uint32_t i = (uint32_t)(KEEP * TAU_THREADS) + (uint32_t)tid; i < n_tiles; i += (uint32_t)TAU_THREADS) {
                const uint32_t v = tm[i];
                const bool match = prefix_bits == 0 || (v >> mshift) == prefix;
                if (match) atomicAdd(&lhist[(v >> shift) & mask], 1u);
            }
            __syncthreads();
            uint32_t bin, above;
            find_bin_256(lhist, need, bin, above, sh);
            need -= above;
            prefix = (prefix << widths[level]) | bin;
            prefix_bits += widths[level];
        }
    }
    if (tid != 0) return;
    const uint32_t tk = prefix > 1u ? prefix : 1u;
    float t = -INFINITY, v = -INFINITY, step = 0.0f;
    const float na = ctl->na, vmax = sqrtf(ctl->vmax_sq);
    const bool in_range = na * vmax < 1e30f && (METRIC != 0 || na > 0.0f);  // false for NaN too
    if (tk >= 0x007FFFFFu && in_range) {
        const float eps = bf16_eps<METRIC>(d, na, vmax, ctl->rho);
        const float tv = key_score(tk);
        if (METRIC == 0) { t = tv / na - 2.02f * eps; v = t * na; step = 2.02f * eps * na; }
        else if (METRIC == 1) { t = (tv - ctl->na_sq) - 2.02f * eps; v = t + ctl->na_sq; step = 2.02f * eps; }
        else { t = tv - 2.02f * eps; v = t; step = 2.02f * eps; }
        step *= 1.0f / (float)DYN_SUB;
    }
    if (!(t == t) || !(v == v)) { t = -INFINITY; v = -INFINITY; }
    if (!(v > -INFINITY && v < INFINITY && step > 0.0f && step < INFINITY)) step = 0.0f;
    ctl->t_key = t > -INFINITY ? score_key(t) : 1u;
    thr[q] = v;
    dyn_lvl[q] = 0u;
    dyn_lad[2 * q] = v;
    dyn_lad[2 * q + 1] = step;
}

// ---------------------------------------------------------------------------
// The GEMM. WGM (queries) x WGN (rows) waves; each wave owns TM x 2
// tiles of v_mfma_f32_32x32x16_bf16, i.e. 32*TM queries x 64 corpus rows.
// MFMA roles as in the f32 kernel: A = queries (accumulator rows), B = corpus
// rows (accumulator columns = lanes). Lane l (r = l & 31, h = l >> 5) feeds
// A[row r][k = 8h..8h+7] and B[k = 8h..8h+7][col r]: one 16-byte LDS read each
// from row-major bf16 images with a 144-byte row stride.
// MODE 0: sample pass, writes per-(query, 64-row tile) maxima as ordered keys.
// MODE 1: full pass, appends rows whose approximate score reaches thr[query].
// METRIC: 0 cosine, 1 Euclidean, 2 dot product.
// ---------------------------------------------------------------------------
// MIR: the corpus operand comes from the bf16 mirror (half the bytes, no conversion);
// the mirror holds exactly the values the conversion below produces, so both forms of
// the pass compute the same accumulators.
// DMA (needs MIR): both operands are bf16 in memory, so nothing has to pass through registers on its
// way to LDS: `global_load_lds_dwordx4` moves 16 bytes per lane straight into LDS (a wave-instruction
// fills 1 KB = 8 rows x 128 bytes), two stages alternate, and a k-step costs ONE barrier: wait for
// this step's data, barrier, start the next step's transfers into the other stage, multiply. The LDS
// rows are unpadded (128 bytes); bank conflicts are avoided by storing 16-byte chunk c of row r at
// slot c ^ ((r >> 1) & 7) -- the permutation is applied to the global address each lane fetches.
template <int METRIC, int TM, int WGM, int WGN, int MODE, bool MIR, bool DMA = false>
__global__ void __launch_bounds__(64 * WGM * WGN)
gemm_bf16_kernel(const float* __restrict__ corpus, const uint16_t* __restrict__ mirror,
                 const __bf16* __restrict__ Qb,
                 const float* __restrict__ norm2, const uint32_t* __restrict__ dead, uint32_t n_rows,
                 uint32_t d, uint32_t n_queries, uint32_t q_tiles, uint32_t tile_step,
                 SearchCtl* __restrict__ ctls, const float* __restrict__ thr,
                 uint32_t* __restrict__ tmax, size_t tmax_stride, uint32_t* __restrict__ cand,
                 uint32_t* __restrict__ cvals, uint32_t need, uint32_t* __restrict__ dyn_lvl,
                 const float* __restrict__ dyn_lad, uint32_t* __restrict__ dyn_cnt) {
    constexpr int NT = 64 * WGM * WGN;  // threads
    constexpr int PF = 1;               // steps of operand loads in flight (2 and 3 measured slower, also with the
                                        // mirror's small 32-register step: 256 queries x 1M x 768 0.634 -> 0.660 ms)
    // (128-column staging steps, i.e. half the barriers, measured 5 % slower with the mirror and spill
    // with f32 rows.)
    // (Two LDS stages with one barrier per k-step were measured too: 256 queries 3 % faster,
    // 32 queries 21 % slower -- one workgroup per CU where three fit with a single stage.)
    // (Two LDS stages with a rotating register stage -- each 16-byte piece written to the other stage
    // a few MFMAs after the last and its registers at once reloaded for the step after next, so that
    // the loads in flight never drop to zero; counted waits, no spills in the loop -- measured slower:
    // full pass over 1M x 768 f32 rows, 256 queries, 0.70 ms against 0.64. The same bytes are in
    // flight either way (one step per thread), and the gap while it stages is not what the pass lacks.)
    // (Neither is it the access pattern or the bytes in flight: a probe that reads 256 bytes from each of
    // the tile's rows per step, with the query slab and the LDS writes beside it, streams at 6.7-7.1 TB/s
    // with ONE step in flight -- scripts/diag_slab.hip. A wave-specialised form was built to find out
    // what it is: 256 queries x 128 rows, four producer waves that only stream the f32 corpus (four
    // k-steps deep in registers -> bf16 -> LDS), four consumer waves that bring the query slab in by
    // direct-to-LDS transfers two steps ahead and issue all the MFMAs, two / three LDS stages, one
    // barrier per step, persistent over the tiles so that nothing drains at a tile's end. Bit-identical
    // results; full pass over 1M x 768, 256 queries: 0.73 ms against this kernel's 0.64-0.66. Taken
    // apart on the same device (no hits listed in any of these): corpus stream + conversion + fragment
    // reads alone 0.47 ms -- HBM's rate --, + the query transfers 0.57, + the MFMAs instead 0.59, both
    // 0.71, + the hits 0.76. The pieces ADD instead of overlapping: every one of them is far from its
    // own limit (matrix pipe 25-45 %, LDS and the vector-memory path about half), but each k-step
    // strings them together through its barrier -- transfers landed, stage written, fragments read,
    // MFMAs issued, by in-order waves -- and what a step cannot hide it pays in full.)
    constexpr int BM = 32 * TM * WGM;   // queries per workgroup tile
    constexpr int BN = 64 * WGN;        // corpus rows per workgroup tile
    constexpr int NLA = BM * 8 / NT;    // 16-byte chunks of the query tile per thread and step
    constexpr int NLB = MIR ? BN * 8 / NT : BN * 16 / NT;  // 16-byte chunks (8 bf16 / 4 f32) of the corpus tile per thread and step
    static_assert(!DMA || MIR, "direct-to-LDS staging needs bf16 operands in memory");
    constexpr int LDS_DWORDS = DMA ? (BM + BN) * 32 : (BM + BN) * BF_LD;
    __shared__ __attribute__((aligned(1024))) uint32_t lds0[LDS_DWORDS];            // (DMA: stage 0)
    __shared__ __attribute__((aligned(1024))) uint32_t lds1[DMA ? LDS_DWORDS : 4];  // DMA: stage 1
    uint32_t* const As = lds0;
    uint32_t* const Bs = lds0 + BM * (DMA ? 32 : BF_LD);
    __shared__ __attribute__((aligned(16))) float qs_s[BM];  // MODE 1: the queries' thresholds
    // ... which tighten during the pass: a workgroup starts from the ladder level the pass has reached
    // by then (trag_device.h: dyn_level_value), and its hits are counted for the levels above it
    __shared__ float qs_lad[MODE == 1 ? BM : 1][2];
    __shared__ uint32_t qs_lvl[MODE == 1 ? BM : 1];
    const int tid = threadIdx.x, lane = tid & 63;
    // (the wave's number as a scalar: what is derived from it -- LDS addresses of the wave's part of a
    // tile, the M0 values of the direct-to-LDS transfers -- then lives in scalar registers)
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WGN, wn = wid % WGN;

    // workgroups sharing a corpus tile (different query tiles) are adjacent on one XCD
    const uint32_t b = blockIdx.x;
    const uint32_t xcd = b & 7u, i8 = b >> 3;
    const uint32_t qt = i8 % q_tiles;
    const uint32_t nts = (i8 / q_tiles) * 8u + xcd;  // index among the tiles of this pass
    const uint32_t nt = nts * tile_step;
    const uint32_t n_row_tiles = (n_rows + BN - 1) / BN;
    if (nt >= n_row_tiles) return;
    const uint32_t q0 = qt * BM, n0 = nt * BN;

    if (MODE == 1 && tid < BM) {
        // (the ladder arrays are padded to a multiple of 256 queries: +inf thresholds behind the batch)
        const uint32_t q = q0 + (uint32_t)tid;
        const uint32_t lv = dyn_lvl[q];
        const float t0 = dyn_lad[2 * q], step = dyn_lad[2 * q + 1];
        qs_lvl[tid] = lv;
        qs_lad[tid][0] = t0;
        qs_lad[tid][1] = step;
        qs_s[tid] = dyn_level_value(lv, t0, step);
    }

    const uint32_t nk = d / BF_K;
    v16f acc[TM][2];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;
    const int frow = lane & 31, half = lane >> 5;

    if constexpr (DMA) {
        constexpr int NW = NT / 64;           // waves
        constexpr int NA = BM / 8 / NW;       // 8-row blocks of the query tile per wave and step
        constexpr int NB = BN / 8 / NW;       // ... of the corpus tile
        const int lrow = lane >> 3, slot = lane & 7;
        const __bf16* asrc[NA];
        const uint16_t* bsrc[NB];
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int row = (wid * NA + i) * 8 + lrow;
            asrc[i] = Qb + (size_t)(q0 + row) * d + ((slot ^ ((row >> 1) & 7)) << 3);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int row = (wid * NB + i) * 8 + lrow;
#ifdef BF_EXP_SAMETILE
            const uint32_t r = (nts % 64u) * (uint32_t)BN + (uint32_t)row;   // (ablation: 8 tiles per XCD, L2-resident; results wrong)
#else
            const uint32_t r = min(n0 + (uint32_t)row, n_rows - 1u);
#endif
            bsrc[i] = mirror + (size_t)r * d + ((slot ^ ((row >> 1) & 7)) << 3);
        }
        typedef __attribute__((address_space(3))) void* lds_ptr;
        typedef const __attribute__((address_space(1))) void* glb_ptr;
#if BF_ASM_XFER
        // the same transfers by inline assembly: a scalar base per 8-row block and k-step plus one 32-bit
        // lane offset fixed for the workgroup's life (no 64-bit vector address arithmetic per transfer), M0
        // set beside it; the compiler does not know them as LDS writes, so nothing it reads from LDS waits
        // for them -- the wait is explicit, in front of the step barrier
        uint32_t voff_a[NA], voff_b[NB];
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int row = (wid * NA + i) * 8 + lrow;
            voff_a[i] = (uint32_t)lrow * d * 2u + (uint32_t)((slot ^ ((row >> 1) & 7)) << 4);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int row = (wid * NB + i) * 8 + lrow;
            const uint32_t r = min(n0 + (uint32_t)row, n_rows - 1u);
            voff_b[i] = (r - n0) * d * 2u + (uint32_t)((slot ^ ((row >> 1) & 7)) << 4);
        }
        [[maybe_unused]] const char* const a_base = reinterpret_cast<const char*>(Qb) + (size_t)(q0 + (uint32_t)(wid * NA * 8)) * d * 2u;
        [[maybe_unused]] const char* const b_base = reinterpret_cast<const char*>(mirror) + (size_t)n0 * d * 2u;
        typedef __attribute__((address_space(3))) uint32_t* lds_u32 [[maybe_unused]];
#endif
        auto transfer = [&](uint32_t* dst, uint32_t kt) __attribute__((always_inline)) {
            uint32_t* b_dst = dst + BM * 32;
#if BF_ASM_XFER && defined(__HIP_DEVICE_COMPILE__)
            const uint32_t m_a = (uint32_t)(uintptr_t)(lds_u32)(dst) + (uint32_t)(wid * NA) * 1024u;
            const uint32_t m_b = (uint32_t)(uintptr_t)(lds_u32)(b_dst) + (uint32_t)(wid * NB) * 1024u;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const char* sb = a_base + (size_t)(i * 8) * d * 2u + (size_t)kt * (BF_K * 2);
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m_a + (uint32_t)i * 1024u), "v"(voff_a[i]), "s"(sb) : "memory");
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const char* sb = b_base + (size_t)kt * (BF_K * 2);
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt" ::"s"(m_b + (uint32_t)i * 1024u), "v"(voff_b[i]), "s"(sb) : "memory");
            }
            return;
#endif
#pragma unroll
            for (int i = 0; i < NA; ++i)
                __builtin_amdgcn_global_load_lds((glb_ptr)(asrc[i] + (size_t)kt * BF_K),
                                                 (lds_ptr)(dst + (wid * NA + i) * 256), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < NB; ++i)
                __builtin_amdgcn_global_load_lds((glb_ptr)(bsrc[i] + (size_t)kt * BF_K),
                                                 (lds_ptr)(b_dst + (wid * NB + i) * 256), 16, 0, 2 /* nt */);
        };
        const int sw = (frow >> 1) & 7;
        // One k-step: start the next step's transfers into `nxt`, multiply out of `cur`. The two
        // stages reach this helper as distinct __restrict__ arguments: the alias scopes that gives
        // the accesses let the compiler see that the fragment reads do not depend on the transfers
        // just issued (otherwise it waits for them before the first read and nothing overlaps).
        auto kstep = [&](const uint32_t* __restrict__ cur, uint32_t* __restrict__ nxt, bool more, uint32_t kt_next)
                         __attribute__((always_inline)) {
#if !BF_ASM_XFER
#ifndef BF_EXP_NOXFER   // (ablation: only the first step's transfers; results wrong)
            if (more) transfer(nxt, kt_next);
#endif
#endif
            const uint32_t* Bc = cur + BM * 32;
            // fragments ping-pong through two register sets: those of MFMA step ks + 1 are read
            // while the MFMAs of step ks run (the staging registers this path does not need pay for it)
            bf16x8 fa[2][TM], fb[2][2];
            auto frags = [&](int buf, int ks) {
                const int chunk = ((ks * 2 + half) ^ sw) << 2;
#pragma unroll
                for (int m = 0; m < TM; ++m)
                    fa[buf][m] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(
                        &cur[(wm * 32 * TM + m * 32 + frow) * 32 + chunk]));
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    fb[buf][n] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(
                        &Bc[(wn * 64 + n * 32 + frow) * 32 + chunk]));
            };
            frags(0, 0);
#if BF_ASM_XFER
            // (the step's first fragment reads are on their way while the transfers are issued. Measured and
            // not kept: the younger wave of each SIMD issuing its transfers in the middle of the step instead,
            // so that the two never stage at once -- 476 us against 449: what it issues late lands late.)
            if (more) transfer(nxt, kt_next);
#endif
#pragma unroll
            for (int ks = 0; ks < BF_K / 16; ++ks) {
#if BF_FRAG_SCHED == 1
                // the next step's fragment reads go out behind this step's first two MFMAs: the first MFMA of
                // a k-step then waits for six reads, not twelve, and the others travel under MFMAs
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1][0], fb[ks & 1][n], acc[0][n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (ks + 1 < BF_K / 16) frags((ks + 1) & 1, ks + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 1; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1][m], fb[ks & 1][n], acc[m][n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#elif BF_FRAG_SCHED == 2
                // ... and spread: two reads behind every pair of MFMAs, the corpus fragments first
                {
                    const int nb = (ks + 1) & 1;
                    const bool more_k = ks + 1 < BF_K / 16;
                    const int chunk = (((ks + 1) * 2 + half) ^ sw) << 2;
#pragma unroll
                    for (int m = 0; m < TM; ++m) {
#pragma unroll
                        for (int n = 0; n < 2; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1][m], fb[ks & 1][n], acc[m][n], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        if (more_k) {
                            if (m == 0) {
#pragma unroll
                                for (int n = 0; n < 2; ++n)
                                    fb[nb][n] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(
                                        &Bc[(wn * 64 + n * 32 + frow) * 32 + chunk]));
                            } else {
                                // (TM query fragments over the TM - 1 remaining gaps)
#pragma unroll
                                for (int mm = 0; mm < TM; ++mm)
                                    if ((mm * (TM - 1)) / TM == m - 1)
                                        fa[nb][mm] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(
                                            &cur[(wm * 32 * TM + mm * 32 + frow) * 32 + chunk]));
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#else
                if (ks + 1 < BF_K / 16) frags((ks + 1) & 1, ks + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1][m], fb[ks & 1][n], acc[m][n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
        };
#if BF_PINGPONG && BF_ASM_XFER
        // An experiment, off by default (-DBF_PINGPONG=1): the two waves of a SIMD (w and w + NW / 2) take
        // turns -- while one issues the MFMAs of half a k-step, the other reads the fragments of its next half
        // from LDS; a workgroup barrier closes every phase, and the waves NW / 2 .. NW - 1 run one phase behind
        // the others. What prompted it (stamps, -DBF_STAMP, 1 M x 768 x 256 queries): left to themselves the two
        // run the same phases at the same times -- both wait for LDS, then both want the matrix pipe, which the
        // older one gets: its 32 MFMAs take 1770 s_memtime ticks of a k-step (a tick is about three quarters of a
        // shader cycle under this load; the MFMAs of both waves need 1540), the younger one finishes at 2800,
        // and the older one waits for it at the barrier (14 000 of a workgroup's 38 000 ticks; the 1 200 - 2 200
        // ahead of the k-loop are not where the time is). Bit-identical results,
        // but SLOWER: 491 us against 449 for the full pass, 50 000 ticks a workgroup. MFMAs are issued
        // asynchronously, so a phase barrier does not mark the end of the multiplying, and a wave's next
        // fragment reads wait for its own MFMAs in flight to release the registers; four barriers a k-step
        // cost more than the alternation gains. (LDS is not the limit either: SQ_LDS_IDX_ACTIVE 17 % of the
        // cycles, no bank conflicts.)
        {
            const bool grp1 = wid >= NW / 2;
            bf16x8 fa[2][TM], fb[2][2];
            auto rd = [&](const uint32_t* cur, int c) __attribute__((always_inline)) {   // MFMA steps 2 c and 2 c + 1
                const uint32_t* Bc = cur + BM * 32;
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    const int chunk = (((2 * c + bb) * 2 + half) ^ sw) << 2;
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        fb[bb][n] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(
                            &Bc[(wn * 64 + n * 32 + frow) * 32 + chunk]));
#pragma unroll
                    for (int m = 0; m < TM; ++m)
                        fa[bb][m] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(
                            &cur[(wm * 32 * TM + m * 32 + frow) * 32 + chunk]));
                }
            };
            auto mm = [&]() __attribute__((always_inline)) {
#pragma unroll
                for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                    for (int m = 0; m < TM; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[bb][m], fb[bb][n], acc[m][n], 0, 0, 0);
            };
            // (LDS only: the transfers in flight are waited for once per k-step, not at every phase)
#define BF_PP_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define BF_PP_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
            static_assert(BF_K == 64, "two halves of two MFMA steps");
            // Every wave runs the same four phases per k-step -- read half 0, multiply, read half 1, multiply --
            // the second group one barrier behind the first. Transfers of step kt + 1 go into the stage that
            // step kt - 1 was read from, free once the second group has read its second half: the first group
            // issues them at the top of its step kt, the second group in the last phase of its step kt - 1 (the
            // same moment), and each waits for its own three phases later, in front of the barrier behind which
            // the first group starts to read them.
#ifdef BF_STAMP
            unsigned long long pp_t = __builtin_amdgcn_s_memtime(), pp_0 = pp_t, pp_rd = 0, pp_mm = 0, pp_vm = 0, pp_bar = 0;
#define BF_PP_T(acc_) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); acc_ += n_ - pp_t; pp_t = n_; }
#else
#define BF_PP_T(acc_) ((void)0)
#endif
            auto step = [&](const uint32_t* cur, uint32_t* nxt, uint32_t kt) __attribute__((always_inline)) {
                if (!grp1 && kt + 1u < nk) transfer(nxt, kt + 1u);
                rd(cur, 0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                BF_PP_T(pp_rd);
                BF_PP_BAR();
                BF_PP_T(pp_bar);
                mm();
                BF_PP_T(pp_mm);
                BF_PP_BAR();
                BF_PP_T(pp_bar);
                rd(cur, 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                BF_PP_T(pp_rd);
                if (grp1) BF_PP_VM();
                BF_PP_T(pp_vm);
                BF_PP_BAR();
                BF_PP_T(pp_bar);
                if (grp1 && kt + 2u < nk) transfer(const_cast<uint32_t*>(cur), kt + 2u);
                mm();
                BF_PP_T(pp_mm);
                if (!grp1) BF_PP_VM();
                BF_PP_T(pp_vm);
                BF_PP_BAR();
                BF_PP_T(pp_bar);
            };
            transfer(lds0, 0u);
            BF_PP_VM();
            BF_PP_BAR();
            if (grp1) {
                if (nk > 1u) transfer(lds1, 1u);
                BF_PP_BAR();
            }
            for (uint32_t kt = 0; kt < nk; kt += 2) {
                step(lds0, lds1, kt);
                if (kt + 1u < nk) step(lds1, lds0, kt + 1u);
            }
            if (!grp1) BF_PP_BAR();
#ifdef BF_STAMP
            if (MODE == 1 && lane == 0 && (blockIdx.x % 997u) == 5u)
                printf("blk %u wave %d ksteps %u read %llu mfma %llu vmwait %llu barrier %llu total %llu\n", blockIdx.x, wid, nk,
                       pp_rd, pp_mm, pp_vm, pp_bar, (unsigned long long)__builtin_amdgcn_s_memtime() - pp_0);
#endif
        }
#else
        transfer(lds0, 0u);
#ifdef BF_EXP_NOBAR
#define BF_STEP_BARRIER() ((void)0)   // (ablation: what the step barrier costs; results wrong)
#elif BF_ASM_XFER
#define BF_STEP_BARRIER() { __builtin_amdgcn_s_waitcnt(0x0F70); __syncthreads(); }   // (vmcnt(0): this wave's transfers)
#else
#define BF_STEP_BARRIER() __syncthreads()
#endif
#ifdef BF_STAMP
        // (where a k-step's time goes: cycles spent waiting at the step barrier -- own transfers, then the
        // other waves -- against the rest, per wave, printed by a few workgroups)
        unsigned long long st_bar = 0, st_work = 0, st_t = __builtin_amdgcn_s_memtime(), st_0 = st_t;
#define BF_STAMP_BAR() { const unsigned long long a_ = __builtin_amdgcn_s_memtime(); st_work += a_ - st_t; __builtin_amdgcn_s_waitcnt(0x0F70); const unsigned long long b0_ = __builtin_amdgcn_s_memtime(); st_vm += b0_ - a_; __syncthreads(); st_t = __builtin_amdgcn_s_memtime(); st_bar += st_t - b0_; }
        unsigned long long st_vm = 0;
#else
#define BF_STAMP_BAR() BF_STEP_BARRIER()
#endif
        for (uint32_t kt = 0; kt < nk; kt += 2) {
            BF_STAMP_BAR();  // this step's transfers have landed (every wave waited for its own before
                             // the barrier) and every wave is done with the other stage
            kstep(lds0, lds1, kt + 1u < nk, kt + 1u);
            if (kt + 1u < nk) {
                BF_STAMP_BAR();
                kstep(lds1, lds0, kt + 2u < nk, kt + 2u);
            }
        }
#ifdef BF_STAMP
        if (MODE == 1 && lane == 0 && (blockIdx.x % 997u) == 5u)
            printf("blk %u wave %d ksteps %u work %llu vmwait %llu barrier %llu total %llu\n", blockIdx.x, wid, nk, st_work,
                   st_vm, st_bar, (unsigned long long)__builtin_amdgcn_s_memtime() - st_0);
#endif
#endif   // BF_PINGPONG
        // (The same staging for the 128 x 128 tile of 65..128 queries measured 12 % slower than its
        // register-staged form, so only the 256 x 256 tile uses it.)
        // (Memory latency is not what this pass lacks either: with the corpus operand taken from a window of
        // 8 tiles per XCD, resident in its L2 -- -DBF_EXP_SAMETILE, results wrong, timing only -- the full
        // pass over 1 M x 768 x 256 queries takes 482 us against 504: deeper prefetch has 4 % to win.)
        // (A persistent form -- one workgroup per CU walking through its tiles, step 0 of the next
        // tile in flight during the epilogue -- measured 6 % slower: the extra live pointers spill.)
    } else {
    const u32x4* aptr[NLA];
    const char* bptr[NLB];
    int a_off[NLA], b_off[NLB];
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
        const int c = tid + NT * i, row = c >> 3, col = c & 7;
        aptr[i] = reinterpret_cast<const u32x4*>(Qb + (size_t)(q0 + row) * d) + col;
        a_off[i] = row * BF_LD + col * 4;
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
        const int c = tid + NT * i, row = MIR ? c >> 3 : c >> 4, col = MIR ? c & 7 : c & 15;
        const uint32_t r = min(n0 + (uint32_t)row, n_rows - 1u);
        if (MIR) {
            bptr[i] = reinterpret_cast<const char*>(mirror + (size_t)r * d + col * 8);
            b_off[i] = row * BF_LD + col * 4;
        } else {
            bptr[i] = reinterpret_cast<const char*>(corpus + (size_t)r * d + col * 4);
            b_off[i] = row * BF_LD + col * 2;
        }
    }
    // PF steps of operand loads are kept in flight in registers: with one or two waves per
    // SIMD a single step in flight leaves the pass bound by HBM latency, not bandwidth
    u32x4 ra[PF][NLA];
    u32x4 rb[PF][NLB];  // MIR: 8 bf16; else the bits of 4 f32
    auto prefetch = [&](int slot, uint32_t kt) {
#pragma unroll
        for (int i = 0; i < NLA; ++i) ra[slot][i] = aptr[i][kt * (BF_K / 8)];
#pragma unroll
        for (int i = 0; i < NLB; ++i)
            rb[slot][i] = __builtin_nontemporal_load(
                reinterpret_cast<const u32x4*>(bptr[i] + (size_t)kt * (MIR ? BF_K * 2 : BF_K * 4)));
    };
    auto stage = [&](int slot) {
#pragma unroll
        for (int i = 0; i < NLA; ++i) *reinterpret_cast<u32x4*>(&As[a_off[i]]) = ra[slot][i];
#pragma unroll
        for (int i = 0; i < NLB; ++i) {
            if (MIR) {
                *reinterpret_cast<u32x4*>(&Bs[b_off[i]]) = rb[slot][i];
            } else {
                const bf16x4 h = __builtin_convertvector(__builtin_bit_cast(v4f, rb[slot][i]), bf16x4);
                *reinterpret_cast<u32x2*>(&Bs[b_off[i]]) = __builtin_bit_cast(u32x2, h);
            }
        }
    };
#pragma unroll
    for (int s = 0; s < PF; ++s)
        if ((uint32_t)s < nk) prefetch(s, (uint32_t)s);

#ifdef BF_STAMP
    unsigned long long sr_bar1 = 0, sr_stage = 0, sr_bar2 = 0, sr_rest = 0, sr_t = __builtin_amdgcn_s_memtime(), sr_0 = sr_t;
#define BF_SR(acc_) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); acc_ += n_ - sr_t; sr_t = n_; }
#else
#define BF_SR(acc_) ((void)0)
#endif
    for (uint32_t kt0 = 0; kt0 < nk; kt0 += PF) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            const uint32_t kt = kt0 + (uint32_t)s;
            if (kt < nk) {  // uniform over the workgroup
                BF_SR(sr_rest);
                __syncthreads();
                BF_SR(sr_bar1);
                stage(s);
                BF_SR(sr_stage);
                __syncthreads();
                BF_SR(sr_bar2);
                // (issuing these loads before the barrier above measured 5 % slower)
                if (kt + PF < nk) prefetch(s, kt + PF);
                // All fragments of a k-step are read before its MFMAs start (scheduling barriers):
                // left alone, the register allocator funnels the query fragments through one
                // register and exposes a full LDS latency before every second MFMA (measured:
                // matrix pipe 21 % busy). Double-buffering the fragments in registers (reading those of
                // step ks + 1 under the MFMAs of ks) measured slower: 256 queries x 1M x 768 0.75 -> 0.91 ms.
#if BF_FRAG_SCHED_R == 1
                // fragments ping-pong through two register sets, the reads of MFMA step ks + 1 spread behind
                // the MFMAs of step ks (two reads per pair of MFMAs, the corpus fragments first)
                if (TM == 4) {
                    bf16x8 fa[2][TM], fb[2][2];
                    auto rd_a = [&](int ks, int m) {
                        return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(
                            &As[(wm * 32 * TM + m * 32 + frow) * BF_LD + ks * 8 + half * 4]));
                    };
                    auto rd_b = [&](int ks, int n) {
                        return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(
                            &Bs[(wn * 64 + n * 32 + frow) * BF_LD + ks * 8 + half * 4]));
                    };
#pragma unroll
                    for (int n = 0; n < 2; ++n) fb[0][n] = rd_b(0, n);
#pragma unroll
                    for (int m = 0; m < TM; ++m) fa[0][m] = rd_a(0, m);
#pragma unroll
                    for (int ks = 0; ks < BF_K / 16; ++ks) {
                        const int nb = (ks + 1) & 1;
                        const bool more_k = ks + 1 < BF_K / 16;
#pragma unroll
                        for (int m = 0; m < TM; ++m) {
#pragma unroll
                            for (int n = 0; n < 2; ++n)
                                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1][m], fb[ks & 1][n], acc[m][n], 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                            if (more_k) {
                                if (m == 0) {
#pragma unroll
                                    for (int n = 0; n < 2; ++n) fb[nb][n] = rd_b(ks + 1, n);
                                } else {
#pragma unroll
                                    for (int mm = 0; mm < TM; ++mm)
                                        if ((mm * (TM - 1)) / TM == m - 1) fa[nb][mm] = rd_a(ks + 1, mm);
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                } else
#endif
#pragma unroll
                for (int ks = 0; ks < BF_K / 16; ++ks) {
                    bf16x8 fa[TM], fb[2];
#pragma unroll
                    for (int m = 0; m < TM; ++m)
                        fa[m] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(
                            &As[(wm * 32 * TM + m * 32 + frow) * BF_LD + ks * 8 + half * 4]));
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        fb[n] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(
                            &Bs[(wn * 64 + n * 32 + frow) * BF_LD + ks * 8 + half * 4]));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < TM; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m], fb[n], acc[m][n], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }

#ifdef BF_STAMP
    BF_SR(sr_rest);
    if (MODE == 1 && lane == 0 && (blockIdx.x % 997u) == 5u)
        printf("blk %u wave %d ksteps %u barrier-in %llu stage %llu barrier-out %llu loads+mfma %llu total %llu\n", blockIdx.x,
               wid, nk, sr_bar1, sr_stage, sr_bar2, sr_rest, (unsigned long long)__builtin_amdgcn_s_memtime() - sr_0);
#endif
    }

    // ---- epilogue. v = the row-dependent part of the score, per accumulator:
    //   cosine dot * inv_nb, dot product dot, Euclidean 2 dot - ||v||^2
    // as one fma(dot, scale, -shift) with per-lane (= per corpus row) constants. Rows that
    // take no part (removed, beyond the corpus) get scale 0 and shift +inf (v = -inf); rows
    // whose norm is outside the bound's range must always be candidates: MODE 1 gives
    // them scale 0 and shift -inf (v = +inf), MODE 0 leaves them out of the sample.
    uint32_t rows_[2];
    float scale[2], nshift[2];
    bool live[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const uint32_t row = n0 + wn * 64 + n * 32 + (uint32_t)frow;
        rows_[n] = row;
        live[n] = row < n_rows && !row_is_dead(dead, row);
        float nb2 = 1.0f;
        bool forced = false;
        if (METRIC != 2) {
            nb2 = row < n_rows ? norm2[row] : 0.0f;
            forced = !(nb2 == 0.0f || (nb2 >= 0x1p-100f && nb2 < INFINITY));
        }
        float sc = METRIC == 0 ? (nb2 > 0.0f ? 1.0f / sqrtf(nb2) : 0.0f) : (METRIC == 1 ? 2.0f : 1.0f);
        float sh = METRIC == 1 ? -nb2 : 0.0f;
        if (forced) { sc = 0.0f; sh = MODE == 1 ? INFINITY : -INFINITY; }
        if (!live[n]) { sc = 0.0f; sh = -INFINITY; }
        scale[n] = sc;
        nshift[n] = sh;
    }
    if (MODE == 0) {
        // per (query, 32 rows): the largest v, as an ordered key. The 32 rows are those of 16
        // lanes x 2 accumulator tiles; the 16-lane maximum costs four DPP steps.
        const uint32_t tile32 = (nts * (uint32_t)(BN / 64) + (uint32_t)wn) * 2u + (uint32_t)((lane >> 4) & 1);
#pragma unroll
        for (int m = 0; m < TM; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qi = wm * 32 * TM + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const uint32_t q = q0 + (uint32_t)qi;
                const float v0 = __builtin_fmaf(acc[m][0][r], scale[0], nshift[0]);
                const float v1 = __builtin_fmaf(acc[m][1][r], scale[1], nshift[1]);
                float best = fmaxf(v0, v1);  // fmaxf drops a NaN operand
                best = fmaxf(best, __int_as_float(__builtin_amdgcn_update_dpp(
                                       __float_as_int(best), __float_as_int(best), 0xB1, 0xF, 0xF, false)));
                best = fmaxf(best, __int_as_float(__builtin_amdgcn_update_dpp(
                                       __float_as_int(best), __float_as_int(best), 0x4E, 0xF, 0xF, false)));
                best = fmaxf(best, __int_as_float(__builtin_amdgcn_update_dpp(
                                       __float_as_int(best), __float_as_int(best), 0x141, 0xF, 0xF, false)));
                best = fmaxf(best, __int_as_float(__builtin_amdgcn_update_dpp(
                                       __float_as_int(best), __float_as_int(best), 0x140, 0xF, 0xF, false)));
                if ((lane & 15) == 0 && q < n_queries)
                    tmax[q * tmax_stride + tile32] = (best > -INFINITY && best < INFINITY) ? score_key(best) : 0u;
            }
        }
    } else {
        // hits are rare (a few hundred rows per query and million rows): first one fma and one
        // compare per accumulator into a per-lane bit mask, then a loop over the set bits
        // (what the listing costs the full pass, measured with thresholds no row reaches: 1 M x 768,
        // 256 queries, 693 -> 665 us from f32 rows, 485 -> 459 from the mirror -- one round trip of
        // returning atomics at the end of a workgroup's life, half hidden by the CU's other workgroup)
        uint32_t wmask[TM];
#pragma unroll
        for (int m = 0; m < TM; ++m) {
            uint32_t w = 0u;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const v4f th = *reinterpret_cast<const v4f*>(&qs_s[wm * 32 * TM + m * 32 + 8 * g + 4 * half]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = g * 4 + e;
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const float v = __builtin_fmaf(acc[m][n][r], scale[n], nshift[n]);
                        w |= !(v < th[e]) ? (1u << (n * 16 + r)) : 0u;  // NaN counts as a hit
                    }
                }
            }
            wmask[m] = w;
        }
#pragma unroll
        for (int m = 0; m < TM; ++m) {
            uint32_t w = wmask[m];
            while (w) {
                const uint32_t bit = (uint32_t)__ffs((int)w) - 1u;
                w &= w - 1u;
                const uint32_t n = bit >> 4, r = bit & 15u;
                const uint32_t q = q0 + (uint32_t)(wm * 32 * TM + m * 32) + (r & 3u) + 8u * (r >> 2) + 4u * (uint32_t)half;
                const bool ok = n ? live[1] : live[0];
                if (ok && q < n_queries) {
                    // the hit's approximate value goes with it (qs_prune_kernel shortens the list with
                    // it before anything is scored exactly); the accumulator is picked by a run-time
                    // index: rare, so the register-indexed move it compiles to costs nothing measurable
                    const float a = n ? acc[m][1][r] : acc[m][0][r];
                    const float v = __builtin_fmaf(a, n ? scale[1] : scale[0], n ? nshift[1] : nshift[0]);
                    // the ladder levels this row vouches for, above the one the workgroup started from:
                    // their counters and the list cursor in one round trip to memory
                    const uint32_t ql = q - q0;
                    const int jh = dyn_hit_level(v, qs_lad[ql][0], qs_lad[ql][1], (int)DYN_LEVELS, (int)DYN_SUB);
                    const int jlo = max((int)qs_lvl[ql], jh - (int)DYN_MAXINC);
                    uint32_t old[DYN_MAXINC];
#pragma unroll
                    for (int i = 0; i < (int)DYN_MAXINC; ++i)
                        old[i] = jh - i > jlo ? atomicAdd(&dyn_cnt[(size_t)q * DYN_LEVELS + (uint32_t)(jh - i)], 1u) : 0xFFFFFFF0u;
                    const uint32_t pos = atomicAdd(&ctls[q].cand_count, 1u);
                    if (pos < CAND_CAP) {
                        cand[(size_t)q * CAND_CAP + pos] = n ? rows_[1] : rows_[0];
                        cvals[(size_t)q * CAND_CAP + pos] = __float_as_uint(v);
                    }
#pragma unroll
                    for (int i = 0; i < (int)DYN_MAXINC; ++i)
                        if (old[i] + 1u == need) atomicMax(&dyn_lvl[q], (uint32_t)(jh - i));
                }
            }
        }
    }
}

bool bf16_supports(uint32_t dim, uint32_t metric) {
    return (dim % BF_K) == 0 && metric <= TRAG_DOT_PRODUCT;
}

// Corpus tiles per sampled tile. The sample must hold many more 32-row groups than
// the rows wanted, or its k-th maximum says little; and it must be dense enough
// that the candidate lists stay short. With a fixed threshold a list would hold about
// need * step * w rows, w = the growth of the rank over a score interval of 2 eps near the
// top of the corpus (for unit-norm rows with independent components the scores have
// sigma = 1 / sqrt(d) and the top sits near 5 sigma: w = exp(10 eps sqrt(d)),
// 4.8 at d = 384, 9.5 at 768, ~390 at 4096). The thresholds tighten during the full pass
// (trag_device.h: dyn_hit_level): after n rows a query's threshold follows the need-th best of
// those n, so the list grows like need * w * (1 + ln step) and a sparse sample costs little.
// Clustered corpora grow faster; their lists overflow and those queries are repeated exactly.
double bf16_rank_growth(uint32_t dim) {
    const double fd = (double)dim;
    const double eps = 0.0078125 * 1.003 + fd * 2.3841858e-07 * 1.01 + (2.0 * fd + 64.0) * 5.9604645e-08;
    return std::exp(10.0 * eps * std::sqrt(fd));
}

// most tiles between two sampled ones for which a list is expected to stay within 1024 rows
double bf16_step_by_list(uint32_t need, uint32_t dim) {
    const double nw = (double)(need ? need : 1) * bf16_rank_growth(dim);
    const double fixed = 1024.0 / nw;                              // thresholds that never move
    const double x = 1024.0 / (1.5 * nw) - 1.0;                    // 1.5: the levels lag behind the rows seen
    const double tight = x > 0.0 ? std::exp(std::min(x, 20.0)) : 1.0;
    return std::max(fixed, tight);
}

// At most every 128th tile is sampled. On a million rows the cap does not matter (1 M x 768, 256 queries, ms per
// step from f32 rows / from the mirror: every 8th tile at most 0.86 / 0.62, 16th 0.82 / 0.61, 32nd 0.81 / 0.61,
// 64th 0.81 / 0.61: there the sample is one round of workgroups whatever its step); on a long corpus the sample
// is bytes -- every 16th tile of 10 M x 384 is a sixteenth of the pass again -- and with the thresholds
// tightening during the full pass a sparse one does as well: 10 M x 384, ms per step with a cap of 16 / 64 / 128,
// 8 queries 2.377 / 2.247 / 2.229 (one query: 2.217), 32 queries 2.398 / 2.259 / 2.244, 128 queries 2.779 / 2.605 /
// 2.586, 32 queries from the mirror 1.258 / 1.177 / 1.168.
#ifndef BF16_MAX_STEP
#define BF16_MAX_STEP 128
#endif
uint32_t bf16_sample_step(uint64_t n_rows, uint32_t need, uint32_t dim) {
    const uint64_t n_tile64 = (n_rows + 63) / 64;
    if (need == 0) need = 1;
    uint64_t step = n_tile64 / (64ull * need);
    const double by_list = bf16_step_by_list(need, dim);
    if ((double)step > by_list) step = (uint64_t)by_list;
    return (uint32_t)(step < 1 ? 1 : (step > BF16_MAX_STEP ? BF16_MAX_STEP : step));
}

// workgroup tile by batch size: queries x corpus rows
static int bf16_variant(uint32_t n_queries) {
    return n_queries <= 32 ? 0 : (n_queries <= 64 ? 1 : (n_queries <= 128 ? 2 : 3));
}
// (The sample pass runs at the full pass's rate per byte, not at a latency floor: with 128 x 128 tiles --
// four times as many workgroups, each a quarter of the work -- the sample of 1 M x 768 for 256 queries,
// every 16th tile, took 67 us instead of 50 from f32 rows and 48 instead of 36 from the mirror.)
static uint32_t bf16_tile_rows(uint32_t n_queries) {
    const int v = bf16_variant(n_queries);
    return v == 2 ? 128 : 256;
}

// thin (sample pass of the 256-query tile only): 256 queries x 64 rows per workgroup -- see bf16_scan_metric
template <int METRIC, int MODE, bool MIR>
static void launch_gemm_bf16(uint32_t n_queries, const ScanArgs& a, const uint16_t* d_mirror, const float* d_norm2,
                             const Bf16Scratch& s, uint32_t tile_step, uint32_t need, hipStream_t st, bool thin = false) {
    const int variant = bf16_variant(n_queries);
    const uint32_t BM = variant == 0 ? 32 : (variant == 1 ? 64 : (variant == 2 ? 128 : 256));
    const uint32_t BN = thin ? 64u : bf16_tile_rows(n_queries);
    const uint32_t q_tiles = (n_queries + BM - 1) / BM;
    const uint32_t n_row_tiles = (a.n_rows + BN - 1) / BN;
    const uint32_t pass_tiles = (n_row_tiles + tile_step - 1) / tile_step;
    const uint32_t grid = ((pass_tiles + 7) / 8) * q_tiles * 8;
#define TRAG_LAUNCH_BF16(TM, WGM, WGN)                                                                       \
    hipLaunchKernelGGL((gemm_bf16_kernel<METRIC, TM, WGM, WGN, MODE, MIR>), dim3(grid), dim3(64 * WGM * WGN), 0, \
                       st, a.corpus, d_mirror, reinterpret_cast<const __bf16*>(s.qb), d_norm2, a.dead, a.n_rows, \
                       a.dim, n_queries, q_tiles, tile_step, s.ctls, s.thr, s.tmax, s.tmax_stride, s.cand, s.ckeys, \
                       need, s.dyn_lvl, s.dyn_lad, s.dyn_cnt)
    if (thin) {
        if constexpr (MODE == 0) TRAG_LAUNCH_BF16(1, 8, 1);   // 256 queries x 64 rows, eight waves of 32 queries
    } else if (variant == 0) TRAG_LAUNCH_BF16(1, 1, 4);       // 32 queries x 256 rows
    else if (variant == 1) TRAG_LAUNCH_BF16(2, 1, 4);  // 64 x 256
    else if (variant == 2) TRAG_LAUNCH_BF16(2, 2, 2);  // 128 x 128
    else if (MIR && !getenv("TRAG_BF16_NODMA")) {      // 256 x 256 from the mirror: direct-to-LDS staging
        if constexpr (MIR)
            hipLaunchKernelGGL((gemm_bf16_kernel<METRIC, 4, 2, 4, MODE, true, true>), dim3(grid), dim3(512), 0, st,
                               a.corpus, d_mirror, reinterpret_cast<const __bf16*>(s.qb), d_norm2, a.dead, a.n_rows,
                               a.dim, n_queries, q_tiles, tile_step, s.ctls, s.thr, s.tmax, s.tmax_stride, s.cand, s.ckeys,
                               need, s.dyn_lvl, s.dyn_lad, s.dyn_cnt);
    } else TRAG_LAUNCH_BF16(4, 2, 4);                  // 256 x 256, eight waves (two per SIMD)
#undef TRAG_LAUNCH_BF16
}

void launch_tau_shift(uint32_t metric, const Bf16Scratch& s, uint32_t n_maxima, uint32_t n_queries, uint32_t padded,
                      uint32_t dim, hipStream_t st);

// The caller has run launch_bf16_prep (norms, control blocks, bf16 queries).
template <int METRIC, bool MIR>
static void bf16_scan_metric(uint32_t n_queries, const ScanArgs& a, const uint16_t* d_mirror,
                             const float* d_norm2, const Bf16Scratch& s, uint32_t need, hipStream_t st) {
    const uint32_t padded = (n_queries + 255u) / 256u * 256u;
    uint32_t step = bf16_sample_step(a.n_rows, need, a.dim);
    // The sample pass of the 256-query tile costs one workgroup-tile's time whatever it samples (every 16th tile of a
    // million rows is one round of workgroups: 53 us, 7 % of the step). With the thresholds tightening during the
    // full pass a thinner sample does as well: the same number of workgroups, each over 64 rows instead of 256
    // (1 M x 768 x 256 queries: sample 51 -> 26 us from f32 rows, 37 -> 21 from the mirror, full pass + 5 us for the
    // longer lists, step 0.765 -> 0.744 / 0.536 -> 0.523 ms; two to four k-steps of loads in flight in the thin
    // kernel change nothing: 26 us is its launch, twelve barrier pairs and the epilogue).
    const bool thin = BF16_THIN_SAMPLE && bf16_variant(n_queries) == 3 && a.n_rows >= 262144u;
    if (thin) step *= 4u;
    launch_gemm_bf16<METRIC, 0, MIR>(n_queries, a, d_mirror, d_norm2, s, step, need, st, thin);
    // 64-row tiles the sample pass wrote maxima for (its last 128/256-row tile may be partial)
    const uint32_t BN = thin ? 64u : bf16_tile_rows(n_queries);
    const uint32_t n_row_tiles = (a.n_rows + BN - 1) / BN;
    const uint32_t pass_tiles = (n_row_tiles + step - 1) / step;
    const uint32_t n_tile64 = pass_tiles * (BN / 32);  // maxima per 32 rows; every sampled workgroup writes all of its own
    launch_tau_shift(METRIC, s, n_tile64, n_queries, padded, a.dim, st);
    launch_gemm_bf16<METRIC, 1, MIR>(n_queries, a, d_mirror, d_norm2, s, 1, need, st);
    // the lists hold every row above the sample's threshold, with its approximate value: keep the
    // rows within 2 eps of the list's need-th largest value (trag_qstat.hip)
    launch_qs_prune(METRIC, s, n_queries, a.dim, st);
}

// n_maxima: tile maxima per query the sample pass wrote (one per 32 sampled rows)
void launch_tau_shift(uint32_t metric, const Bf16Scratch& s, uint32_t n_maxima, uint32_t n_queries, uint32_t padded,
                      uint32_t dim, hipStream_t st) {
    const dim3 grid(n_queries + (padded - n_queries + TAU_THREADS - 1) / TAU_THREADS), block(TAU_THREADS);
#define TRAG_TAU(M) hipLaunchKernelGGL(tau_shift_kernel<M>, grid, block, 0, st, s.tmax, s.tmax_stride, n_maxima, s.ctls, \
                                       n_queries, padded, dim, s.thr, s.dyn_lvl, s.dyn_lad, s.dyn_cnt)
    if (metric == TRAG_COSINE) TRAG_TAU(0);
    else if (metric == TRAG_EUCLIDEAN) TRAG_TAU(1);
    else TRAG_TAU(2);
#undef TRAG_TAU
}

// Which of the two forms serves a batch. The query-stationary form (trag_qstat.hip) reads the
// corpus once for up to 2048 queries, and wins where its workgroups are full: rows of up to 384
// components and more than 128 queries (measured on 1.25 M x 384: 0.49 against 0.51 ms at 256
// queries from the mirror, 0.55 against 0.66 from the f32 rows, 1.46 / 1.65 against 1.90 / 2.41 ms
// at 1000). Smaller batches would pad its 256-query groups, and longer rows leave its waves too
// few registers (built for 768 components it ran 1 M x 768 x 256 in 1.15 / 0.89 ms against the
// tiled kernels' 0.98 / 0.76): the tiled kernels of this file serve those, 256 queries a pass.
// TRAG_BF16_TILED=1 forces the tiled kernels everywhere, for comparison.
bool bf16_uses_qs(uint32_t dim, uint32_t metric, uint32_t n_queries) {
    static int tiled = -1;
    if (tiled < 0) {
        const char* t = getenv("TRAG_BF16_TILED");
        tiled = (t && t[0] == '1') ? 1 : 0;
    }
    return !tiled && qs_supports(dim, metric) && n_queries > 128;
}

uint32_t bf16_max_batch(uint32_t dim, uint32_t metric, uint32_t n_queries) {
    return bf16_uses_qs(dim, metric, n_queries) ? qs_max_batch(dim) : BATCH_QUERIES;
}

// d_mirror != nullptr: the passes read the bf16 mirror instead of the f32 corpus.
hipError_t launch_bf16_scan(const float* d_Q, uint32_t n_queries, const ScanArgs& a, const uint16_t* d_mirror,
                            const float* d_norm2, const Bf16Scratch& s, uint32_t need, uint32_t k,
                            const float* d_vmax_sq, hipStream_t st) {
    const bool qs = bf16_uses_qs(a.dim, a.metric, n_queries);
    const uint32_t group = qs ? qs_group_queries(a.dim) : 256u;
    hipError_t e = launch_bf16_prep(d_Q, n_queries, (n_queries + group - 1) / group * group, a, d_norm2, s, need, k,
                                    d_vmax_sq, qs, st);
    if (e != hipSuccess) return e;
    if (qs) return launch_qs_scan(n_queries, a, d_mirror, d_norm2, s, need, st);
    if (d_mirror) {
        if (a.metric == TRAG_COSINE) bf16_scan_metric<0, true>(n_queries, a, d_mirror, d_norm2, s, need, st);
        else if (a.metric == TRAG_EUCLIDEAN) bf16_scan_metric<1, true>(n_queries, a, d_mirror, d_norm2, s, need, st);
        else bf16_scan_metric<2, true>(n_queries, a, d_mirror, d_norm2, s, need, st);
    } else {
        if (a.metric == TRAG_COSINE) bf16_scan_metric<0, false>(n_queries, a, d_mirror, d_norm2, s, need, st);
        else if (a.metric == TRAG_EUCLIDEAN) bf16_scan_metric<1, false>(n_queries, a, d_mirror, d_norm2, s, need, st);
        else bf16_scan_metric<2, false>(n_queries, a, d_mirror, d_norm2, s, need, st);
    }
    return hipGetLastError();
}

// Rows a query's candidate list would hold, with a threshold that never moved and a sample of at
// most every 8th tile, on a corpus of independent components; AUTO avoids the prefilter when
// it nears the list size.
double bf16_list_estimate(uint64_t n_rows, uint32_t need, uint32_t dim) {
    if (need == 0) need = 1;
    const double w = bf16_rank_growth(dim);
    double step = (double)(((n_rows + 63) / 64) / (64ull * need));
    step = std::min(step, 1024.0 / ((double)need * w));
    step = step < 1.0 ? 1.0 : (step > 8.0 ? 8.0 : std::floor(step));
    return (double)need * step * w;
}

}  // namespace trag
