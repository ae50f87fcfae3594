// EXPERIMENT, NOT PART OF THE LIBRARY (not built by the Makefile, nothing links it). Kept with its
// measurements because it settles a design question: can the full pass of the bf16 prefilter do
// without the workgroup's tile barrier by keeping the QUERIES in LDS and each wave's corpus rows in
// registers? Measured on MI355X, 1.25 M x 384 x 1000 queries over the bf16 mirror, cycles of one XCD
// (GRBM_GUI_ACTIVE / 8; 937 k = the MFMAs alone), beside the query-stationary kernel that stays:
//   query-stationary (trag_qstat.hip), everything            1 769 k   matrix pipe 53 % busy
//   this kernel, rows loaded in MFMA fragment layout from the
//     row-major mirror (32 rows x 32 bytes per instruction)   2 254 k   41 %  -- 64 cache lines of 16 bytes per
//                                                                        wave-instruction: the vector cache's
//                                                                        tag rate, one line a clock, is the limit
//   ... no hits listed                                        2 026 k
//   ... no hits, no row loads (MFMA + LDS reads + test)       1 208 k   78 %
//   rows read as 1 KB contiguous per instruction (what a mirror
//     stored in fragment order would give; wrong results)     1 633 k   57 %
//   ... no hits                                               1 353 k   69 %
// So even with a second copy of the corpus in fragment order it would be 8 % faster than the kernel in
// use -- listing and flushing hits costs a lone wave per SIMD 280 k cycles -- and it cannot serve the
// f32 rows at all (converted fragments must live in vector registers: two sets do not fit). It also
// has an unresolved parity failure at d = 128 with 200 queries. Build it for timing with
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I include -I trueno-rag_amd/csrc -c <this file>
// after re-adding the two declarations below to trag_internal.h.
//   bool rs_supports(uint32_t dim, uint32_t metric);
//   hipError_t launch_rs_full_pass(uint32_t n_queries, const ScanArgs& a, const uint16_t* d_mirror,
//                                  const Bf16Scratch& s, uint32_t need, hipStream_t st);
//
// Query batches, bf16 matrix-core prefilter over the bf16 mirror, ROW-STATIONARY form of the full pass.
//
// The query-stationary form (trag_qstat.hip) keeps the queries in registers and streams the corpus
// through LDS: every wave of a workgroup reads every corpus tile, the workgroup stages it
// together, and a barrier per tile makes all waves pay for the slowest one -- the wave that happens
// to list a hit, or the one that moves the lists to memory. Measured on 1.25 M x 384 x 1000 queries
// (cycles of one XCD, ideal 937 k for the MFMAs alone): bare MFMA + fragment reads 1 163 k, staging
// + 280 k, listing + flushing the hits + 192 k.
//
// Here the roles are swapped. The QUERIES of a workgroup sit in LDS for the whole kernel (a padded
// bf16 image, written once, then only read); every WAVE owns 32 corpus rows at a time and holds
// them in registers as the B fragments of v_mfma_f32_32x32x16_bf16 over the full K extent, loaded
// straight from the mirror in fragment layout (lane (r, h) loads the 16 bytes [32 ks + 16 h, +16) of
// row r): no staging through LDS, no LDS writes, NO BARRIER in the loop. A wave multiplies its 32
// rows with the workgroup's query blocks one after the other (A fragments from LDS: 1 KB per MFMA
// and wave, half the LDS's rate at four waves), tests a block's 16 accumulators per lane against
// the queries' thresholds between the MFMAs of the next block, and lists hits into its own list;
// meanwhile the 32 rows of its next step arrive in the other half of the fragment registers. Waves
// never wait for one another: a hit costs the wave that has it, nobody else.
//   * one wave per SIMD, 512 registers: 2 x (d / 4) accumulation registers hold the two row sets
//     (the MFMA reads its B operand there directly), the vector registers the accumulators, the
//     A-fragment ring and the thresholds;
//   * the fragment loads are issued by inline assembly into the accumulation registers (a load the
//     compiler emits lands in a vector register, and 96 copies a step would cost more than the
//     loads) and waited for by ONE s_waitcnt vmcnt(0) per step, where they have been in flight for
//     three quarters of a step;
//   * the MFMAs are inline assembly too (the compiler's own put the accumulators into accumulation
//     registers and copy them out for the threshold test). The one hazard the hardware does not
//     interlock -- a vector instruction reading an MFMA's result too early -- is kept away by
//     construction: a block's accumulators become visible to the test three MFMAs (96 cycles of
//     the matrix pipe) after the block's last one;
//   * hits leave the wave without a stall: at the end of a step the lanes issue the atomics that
//     reserve list positions and count the ladder levels (trag_device.h: dyn_hit_level), their
//     results are used behind the next step's s_waitcnt, which has to be there anyway.
// The bound, the sample pass, the thresholds and the exact f32 finish are those of trag_bf16.hip /
// trag_qstat.hip; this file replaces only the full pass when the store has a bf16 mirror.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "trag_device.h"
#include "trag_internal.h"

#ifndef RS_FLUSH_EVERY
#define RS_FLUSH_EVERY 16   // steps after which a wave refreshes the thresholds even without a hit of its own
#endif

namespace trag {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// f(integral_constant<int, I>) for I = FIRST .. LAST - 1: loop indices that must be constants of the
// front end (immediate operands of inline assembly)
template <int FIRST, int LAST, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (FIRST < LAST) {
        f(std::integral_constant<int, FIRST>{});
        static_for<FIRST + 1, LAST>(f);
    }
}

#if defined(__HIP_DEVICE_COMPILE__)
// 16 bytes per lane from base (scalar) + voff + OFF straight into accumulation registers; not known
// to the compiler as a load: waited for by rs_wait_loads.
template <int OFF>
__device__ __forceinline__ void rs_load(u32x4& dst, uint32_t voff, const char* base) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=a"(dst) : "v"(voff), "s"(base), "n"(OFF) : "memory");
}
__device__ __forceinline__ void rs_load_rc(u32x2& dst, uint32_t voff, const char* base) {
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base) : "memory");
}
// every vector-memory operation of the wave has returned; rc passes through so that its readers stay behind
__device__ __forceinline__ void rs_wait_loads(u32x2& rc) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(rc) : : "memory"); }
__device__ __forceinline__ void rs_mfma_first(v16f& acc, const u32x4& a, const u32x4& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(b));
}
__device__ __forceinline__ void rs_mfma(v16f& acc, const u32x4& a, const u32x4& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
}
// readers of acc stay behind this point of the instruction stream
__device__ __forceinline__ void rs_visible(v16f& acc) { asm volatile("" : "+v"(acc)); }
__device__ __forceinline__ void rs_visible_after_nops(v16f& acc) { asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(acc)); }
__device__ __forceinline__ void rs_pin(float x) { asm volatile("" ::"v"(x)); }
#endif

template <int KS, int NQB>
__global__ void __launch_bounds__(256, 1)
rs_gemm_kernel(const uint16_t* __restrict__ mirror, const uint16_t* __restrict__ Qb, const float* __restrict__ rowc_g,
               uint32_t n_rows, uint32_t n_queries, uint32_t G, uint32_t S, SearchCtl* __restrict__ ctls,
               uint32_t* __restrict__ cand, uint32_t* __restrict__ cvals, uint32_t need,
               uint32_t* __restrict__ dyn_lvl, const float* __restrict__ dyn_lad, uint32_t* __restrict__ dyn_cnt) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int D = KS * 16;
    constexpr int STRIDE = D / 2 + 4;     // dwords per image row: 16-byte fragment reads of 16 rows hit 16 bank groups
    constexpr int QN = 32 * NQB;          // queries of a workgroup
    constexpr int NW = 4;                 // waves: one per SIMD
#ifndef RS_NF
#define RS_NF 4
#endif
    constexpr int NF = RS_NF;             // A fragments in flight
    constexpr int HITS = 64;              // capacity of a wave's hit list
    constexpr int NT = KS - 4;            // k-steps of a block that carry a piece of the previous block's test
    constexpr int CH = (16 + NT - 1) / NT;
    static_assert(NQB % 2 == 0 && KS % NF == 0 && KS >= 8 && QN % 64 == 0, "geometry");
    __shared__ __attribute__((aligned(16))) uint32_t img[QN * STRIDE];
    __shared__ __attribute__((aligned(16))) float thr_lds[QN];
    __shared__ __attribute__((aligned(16))) uint32_t lvl_lds[QN];
    __shared__ __attribute__((aligned(16))) float lad_lds[QN][2];
    __shared__ uint32_t hit_q[NW][HITS], hit_row[NW][HITS], hit_v[NW][HITS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    // workgroups of one XCD: G query groups x S row streams (the G readers of a row share an L2)
    const uint32_t b = blockIdx.x;
    const uint32_t xcd = b & 7u, slot = b >> 3;
    const uint32_t g = slot % G;
    const uint32_t strm = (slot / G) * 8u + xcd;
    const uint32_t NS = S * 8u * NW;                       // wave streams per query group
    const uint32_t n_sub = (n_rows + 31u) / 32u;           // 32-row steps of the corpus
    const uint32_t qg0 = g * (uint32_t)QN;                 // first query of the workgroup

    // ---- the queries' image and threshold table ----
    {
        constexpr int PPR = D / 8;                         // 16-byte pieces per query
        const u32x4* src = reinterpret_cast<const u32x4*>(Qb + (size_t)qg0 * D);
        for (int p = tid; p < QN * PPR; p += 256) {
            const int row = p / PPR, c = p % PPR;
            *reinterpret_cast<u32x4*>(&img[row * STRIDE + c * 4]) = src[p];
        }
        if (tid < QN) {
            const uint32_t q = qg0 + (uint32_t)tid;        // (the ladder arrays are padded: +inf behind the batch)
            const uint32_t lv = dyn_lvl[q];
            float t0 = dyn_lad[2 * q];
            const float step = dyn_lad[2 * q + 1];
#ifdef RS_EXP_NOHIT
            t0 = INFINITY;
#endif
            if (t0 == -INFINITY) {                         // outside the bound's range: marked as overflowed, once
                if (strm == 0u) atomicMax(&ctls[q].cand_count, CAND_CAP + 1u);
                t0 = INFINITY;
            }
            lvl_lds[tid] = lv;
            lad_lds[tid][0] = t0;
            lad_lds[tid][1] = step;
            thr_lds[tid] = dyn_level_value(lv, t0, step);
        }
    }
    __syncthreads();

    // ---- per-lane constants ----
#ifdef RS_EXP_COALESCED
    // (timing experiment: the loads of a step as they would be from a mirror stored in fragment order --
    // 1 KB contiguous per instruction; the data are the step's own 24 KB in another order: results are wrong)
    const uint32_t voff = (uint32_t)lane * 16u;
#define RS_KOFF(ks) (((ks) % 4) * 1024)
#define RS_KBASE(ks) (((ks) / 4) * 4096)
#else
    const uint32_t voff = (uint32_t)r * (uint32_t)(D * 2) + (uint32_t)h * 16u;   // byte offset of (row r, half h) in a step's rows
#define RS_KOFF(ks) ((ks) * 32)
#define RS_KBASE(ks) 0
#endif
    const uint32_t rc_off = (uint32_t)r * 8u;
    uint32_t qmask[NQB];                                   // bit e: accumulator e of the block is a query of the batch
#pragma unroll
    for (int j = 0; j < NQB; ++j) {
        qmask[j] = 0u;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const uint32_t q = qg0 + 32u * (uint32_t)j + (uint32_t)((e & 3) + 8 * (e >> 2) + 4 * h);
            qmask[j] |= q < n_queries ? (1u << e) : 0u;
        }
    }
    auto frag = [&](int t) __attribute__((always_inline)) {   // A fragment of linear k-step t of a step (block t / KS)
        const int j = (t / KS) % NQB, ks = t % KS;
        return *reinterpret_cast<const u32x4*>(&img[(j * 32 + r) * STRIDE + 4 * h + ks * 8]);
    };
    float th[NQB][16];
    auto load_thresholds = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NQB; ++j)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const v4f x = *reinterpret_cast<const v4f*>(&thr_lds[32 * j + 8 * g4 + 4 * h]);
#pragma unroll
                for (int e = 0; e < 4; ++e) th[j][4 * g4 + e] = x[e];
            }
    };

    // ---- hits: the wave's own list, moved to memory in two halves (issue / complete) ----
    uint32_t hits_here = 0u;       // entries in the list (wave-uniform)
    uint32_t pending = 0u;         // entries whose atomics are in flight
    uint32_t p_q = 0u, p_row = 0u, p_v = 0u, p_pos = CAND_CAP, p_old[DYN_MAXINC];
    int p_jh = -1;
    uint32_t p_lv[QN / 64];
    bool p_refresh = false;
#pragma unroll
    for (int i = 0; i < (int)DYN_MAXINC; ++i) p_old[i] = 0xFFFFFFF0u;
    auto flush_issue = [&]() __attribute__((always_inline)) {
        // every lane below hits_here owns one entry: the list cursor of its query and the counters of the
        // ladder levels the row vouches for above the level the table knows -- all atomics with a
        // return, none waited for here -- and the current levels of the workgroup's queries
        const bool valid = (uint32_t)lane < hits_here;
        p_q = valid ? hit_q[wid][lane] : qg0;
        p_row = hit_row[wid][lane];
        p_v = hit_v[wid][lane];
        const uint32_t ql = p_q - qg0;
        p_jh = valid ? dyn_hit_level(__uint_as_float(p_v), lad_lds[ql][0], lad_lds[ql][1], (int)DYN_LEVELS, (int)DYN_SUB) : -1;
        const int jlo = max((int)lvl_lds[ql], p_jh - (int)DYN_MAXINC);
#pragma unroll
        for (int i = 0; i < (int)DYN_MAXINC; ++i)
            p_old[i] = (valid && p_jh - i > jlo) ? atomicAdd(&dyn_cnt[(size_t)p_q * DYN_LEVELS + (uint32_t)(p_jh - i)], 1u) : 0xFFFFFFF0u;
        p_pos = valid ? atomicAdd(&ctls[p_q].cand_count, 1u) : CAND_CAP;
#pragma unroll
        for (int t = 0; t < QN / 64; ++t) p_lv[t] = dyn_lvl[qg0 + (uint32_t)(lane + 64 * t)];
        pending = hits_here;
        p_refresh = true;
        hits_here = 0u;
    };
    auto flush_complete = [&]() __attribute__((always_inline)) {
        if (p_pos < CAND_CAP) {
            cand[(size_t)p_q * CAND_CAP + p_pos] = p_row;
            cvals[(size_t)p_q * CAND_CAP + p_pos] = p_v;
        }
#pragma unroll
        for (int i = 0; i < (int)DYN_MAXINC; ++i)
            if (p_old[i] + 1u == need) atomicMax(&dyn_lvl[p_q], (uint32_t)(p_jh - i));
#pragma unroll
        for (int t = 0; t < QN / 64; ++t) {
            const int q = lane + 64 * t;
            thr_lds[q] = dyn_level_value(p_lv[t], lad_lds[q][0], lad_lds[q][1]);
            lvl_lds[q] = p_lv[t];
        }
        pending = 0u;
        p_refresh = false;
        p_pos = CAND_CAP;
#pragma unroll
        for (int i = 0; i < (int)DYN_MAXINC; ++i) p_old[i] = 0xFFFFFFF0u;
    };
    // The previous block's hits: lanes form their hit bits without branches, the lanes that have one write
    // it at consecutive positions of the wave's list (positions by ballot: no atomic). A list that would
    // overflow is moved to memory on the spot (a row that scores high for most queries at once).
    auto list_hits = [&](const v16f& acc, float sc, float sh, const float (&t16)[16], const float (&m)[4], uint32_t row0,
                         uint32_t qblk0, uint32_t mask) __attribute__((always_inline)) {
        const float mall = fmaxf(fmaxf(m[0], m[1]), fmaxf(m[2], m[3]));
        if (__builtin_amdgcn_ballot_w64(!(mall + sh < 0.0f)) != 0ull) {
            const bool live = sh != -INFINITY || sc != 0.0f;
            uint32_t w = 0u;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                if (__builtin_amdgcn_ballot_w64(!(m[gq] + sh < 0.0f)) != 0ull) {
#pragma unroll
                    for (int e = 4 * gq; e < 4 * gq + 4; ++e)
                        w |= !(__builtin_fmaf(acc[e], sc, -t16[e]) + sh < 0.0f) ? (1u << e) : 0u;
                }
            }
            w = live ? (w & mask) : 0u;
            const uint32_t row = row0 + (uint32_t)r;
            const uint32_t qb = qblk0 + 4u * (uint32_t)h;
            unsigned long long any = __builtin_amdgcn_ballot_w64(w != 0u);
            while (any != 0ull) {
                const bool act = w != 0u;
                const uint32_t bit = act ? (uint32_t)__builtin_ctz(w) : 0xFFFFFFFFu;
                w &= w - 1u;
                float a = 0.0f;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    if (__builtin_amdgcn_ballot_w64((bit >> 2) == (uint32_t)gq) != 0ull) {
#pragma unroll
                        for (int e = 4 * gq; e < 4 * gq + 4; ++e) a = bit == (uint32_t)e ? acc[e] : a;
                    }
                }
                const float v = __builtin_fmaf(a, sc, sh);
                const uint32_t add = (uint32_t)__builtin_popcountll(any);
                if (hits_here + add > (uint32_t)HITS) {
                    // (rare) the list is full: finish what is in flight, then move it out and wait
                    if (pending | (uint32_t)p_refresh) flush_complete();
                    flush_issue();
                    flush_complete();
                }
                const uint32_t pos = hits_here + __builtin_amdgcn_mbcnt_hi((uint32_t)(any >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)any, 0u));
                if (act) {
                    hit_q[wid][pos] = qb + (bit & 3u) + 8u * (bit >> 2);
                    hit_row[wid][pos] = row;
                    hit_v[wid][pos] = __float_as_uint(v);
                }
                hits_here += add;
                any = __builtin_amdgcn_ballot_w64(w != 0u);
            }
        }
    };

    // ---- the row fragments: two sets in accumulation registers ----
    u32x4 B0[KS], B1[KS];
    u32x2 rc0 = {0u, 0xFF800000u}, rc1 = {0u, 0xFF800000u};   // (scale 0, shift -inf): "takes no part"
    const char* const mbase = reinterpret_cast<const char*>(mirror);
    const char* const rbase = reinterpret_cast<const char*>(rowc_g);
    // accumulators: two sets that alternate by block (a block's set is tested while the next block
    // accumulates into the other one)
    v16f accA, accB;
#pragma unroll
    for (int e = 0; e < 16; ++e) { accA[e] = 0.0f; accB[e] = 0.0f; }
    float scP = 0.0f, shP = -INFINITY;     // the row constants of the step whose last block is still to be tested
    uint32_t row0P = 0u;
    u32x4 fr[NF];
    uint32_t since_refresh = 0u;

    uint32_t u = strm * (uint32_t)NW + (uint32_t)wid;      // this wave's stream: steps u, u + NS, ...
    {
        const uint32_t uu = u < n_sub ? u : 0u;
        const char* mb = mbase + (size_t)uu * (size_t)(32 * D * 2);
        const char* rb = rbase + (size_t)uu * 256u;
        static_for<0, KS>([&](auto ksc) __attribute__((always_inline)) {
            constexpr int ks = decltype(ksc)::value;
            rs_load<RS_KOFF(ks)>(B0[ks], voff, mb + RS_KBASE(ks));
        });
        rs_load_rc(rc0, rc_off, rb);
    }
#pragma unroll
    for (int t = 0; t < NF; ++t) fr[t] = frag(t);

    // One step = 32 rows x the workgroup's NQB query blocks. CUR: which register set holds the step's rows.
    auto step = [&](auto cur_tag, uint32_t u_cur, uint32_t u_next) __attribute__((always_inline)) {
        constexpr int CUR = decltype(cur_tag)::value;
        // everything this wave has in flight: the step's rows (issued a step ago) and a flush's atomics
        if (CUR == 0) rs_wait_loads(rc0);
        else rs_wait_loads(rc1);
        if (pending | (uint32_t)p_refresh) flush_complete();
        load_thresholds();
        const float sc = __uint_as_float(CUR == 0 ? rc0[0] : rc1[0]);
        const float sh = __uint_as_float(CUR == 0 ? rc0[1] : rc1[1]);
        const uint32_t row0 = u_cur * 32u;
        const uint32_t un = u_next < n_sub ? u_next : 0u;
        const char* mb = mbase + (size_t)un * (size_t)(32 * D * 2);
        const char* rb = rbase + (size_t)un * 256u;
        static_for<0, NQB>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            // block j accumulates into accA (j even) / accB (j odd); the other set holds block j - 1 (for
            // j = 0: the previous step's last block), tested between this block's MFMAs
            float mp[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            const float scT = j == 0 ? scP : sc, shT = j == 0 ? shP : sh;
            const int jp = (j + NQB - 1) % NQB;
            static_for<0, KS>([&](auto ksc) __attribute__((always_inline)) {
                constexpr int ks = decltype(ksc)::value;
                constexpr int t = j * KS + ks;
                {
                    v16f& acc = (j & 1) ? accB : accA;
                    const u32x4& bf = CUR == 0 ? B0[ks] : B1[ks];
                    if (ks == 0) rs_mfma_first(acc, fr[ks % NF], bf);
                    else rs_mfma(acc, fr[ks % NF], bf);
                }
                fr[ks % NF] = frag(t + NF);
#ifndef RS_EXP_NOLOAD
                if (j == 0) {   // the next step's rows, one load behind each MFMA of the first block
                    if (CUR == 0) rs_load<RS_KOFF(ks)>(B1[ks], voff, mb + RS_KBASE(ks));
                    else rs_load<RS_KOFF(ks)>(B0[ks], voff, mb + RS_KBASE(ks));
                    if (ks == 0) {
                        if (CUR == 0) rs_load_rc(rc1, rc_off, rb);
                        else rs_load_rc(rc0, rc_off, rb);
                    }
                }
#endif
                if (ks == 2) {
                    // the previous block's accumulators may be read from here on: three MFMAs of this block
                    // have been issued behind its last one
                    if (j & 1) rs_visible(accA);
                    else rs_visible(accB);
                }
                if constexpr (ks >= 3 && (ks - 3) * CH < 16) {
                    const v16f& accT = (j & 1) ? accA : accB;
#pragma unroll
                    for (int e = (ks - 3) * CH; e < (ks - 2) * CH && e < 16; ++e)
                        mp[e >> 2] = fmaxf(mp[e >> 2], __builtin_fmaf(accT[e], scT, -th[jp][e]));
                    rs_pin(mp[((ks - 3) * CH) >> 2]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ks == KS - 2) {
                    static_assert((KS - 2 - 3) * CH >= 16 - CH, "the test is complete before the hits are listed");
                    const v16f& accT = (j & 1) ? accA : accB;
                    list_hits(accT, scT, shT, th[jp], mp, j == 0 ? row0P : row0, qg0 + 32u * (uint32_t)jp, qmask[jp]);
                }
            });
        });
        scP = sc;
        shP = sh;
        row0P = row0;
        since_refresh += 1u;
        if (hits_here != 0u || since_refresh >= (uint32_t)RS_FLUSH_EVERY) {
            flush_issue();
            since_refresh = 0u;
        }
    };

    while (u < n_sub) {
        step(std::integral_constant<int, 0>{}, u, u + NS);
        u += NS;
        if (u >= n_sub) break;
        step(std::integral_constant<int, 1>{}, u, u + NS);
        u += NS;
    }
    // the last block of the last step, then whatever is still in the list
    rs_wait_loads(rc0);
    if (pending | (uint32_t)p_refresh) flush_complete();
    {
        rs_visible_after_nops(accB);
        load_thresholds();
        float mp[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int e = 0; e < 16; ++e) mp[e >> 2] = fmaxf(mp[e >> 2], __builtin_fmaf(accB[e], scP, -th[NQB - 1][e]));
        list_hits(accB, scP, shP, th[NQB - 1], mp, row0P, qg0 + 32u * (uint32_t)(NQB - 1), qmask[NQB - 1]);
    }
    if (hits_here != 0u) {
        flush_issue();
        flush_complete();
    }
#endif
}

bool rs_supports(uint32_t dim, uint32_t metric);
hipError_t launch_rs_full_pass(uint32_t n_queries, const ScanArgs& a, const uint16_t* d_mirror, const Bf16Scratch& s,
                               uint32_t need, hipStream_t st);

bool rs_supports(uint32_t dim, uint32_t metric) {
    static int off = -1;
    if (off < 0) {
        const char* e = getenv("TRAG_RS");
        off = (e && e[0] == '0') ? 1 : 0;
    }
    return !off && metric <= TRAG_DOT_PRODUCT && (dim == 128 || dim == 256 || dim == 384);
}

template <int KS, int NQB>
static void rs_launch(uint32_t n_queries, const ScanArgs& a, const uint16_t* d_mirror, const Bf16Scratch& s, uint32_t need,
                      hipStream_t st) {
    static int cu_per_xcd = 0;
    if (cu_per_xcd == 0) {
        int dev = 0, n_cu = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
        cu_per_xcd = std::max(1, n_cu / 8);
    }
    const uint32_t QN = 32u * NQB;
    const uint32_t G = (n_queries + QN - 1) / QN;
    const uint32_t S = std::max<uint32_t>(1u, (uint32_t)cu_per_xcd / G);
    hipLaunchKernelGGL((rs_gemm_kernel<KS, NQB>), dim3(8u * S * G), dim3(256), 0, st, d_mirror, s.qb, s.rowc, a.n_rows,
                       n_queries, G, S, s.ctls, s.cand, s.ckeys, need, s.dyn_lvl, s.dyn_lad, s.dyn_cnt);
}

// The full pass of the prefilter over the bf16 mirror (the caller has run the sample pass and the
// thresholds; queries padded to a multiple of 256 as for the query-stationary form).
hipError_t launch_rs_full_pass(uint32_t n_queries, const ScanArgs& a, const uint16_t* d_mirror, const Bf16Scratch& s,
                               uint32_t need, hipStream_t st) {
    switch (a.dim) {
    case 128: rs_launch<8, 4>(n_queries, a, d_mirror, s, need, st); break;
    case 256: rs_launch<16, 4>(n_queries, a, d_mirror, s, need, st); break;
    case 384: rs_launch<24, 4>(n_queries, a, d_mirror, s, need, st); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace trag
