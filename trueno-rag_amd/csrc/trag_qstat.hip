// Query batches, bf16 matrix-core prefilter, QUERY-STATIONARY form.
//
// Q x D^T with N ~ 10^6 corpus rows, B <= 2048 queries and K = d <= 384 is not a square GEMM:
// the corpus is the only large operand and must be read ONCE per batch, whatever B is. So the
// queries stay put and the corpus streams:
//   * a workgroup owns 32 NW QB queries for its whole life; every wave keeps the bf16 fragments
//     of its 32 QB queries over the full K extent in registers (QB d / 4 VGPRs per lane);
//   * the corpus flows through LDS in tiles of RT rows x the whole K extent: all waves of the
//     workgroup load a tile's rows with coalesced 16-byte loads (a step ahead, in registers),
//     convert f32 -> bf16 on the way and write a padded row-major image (two tiles alternate,
//     one barrier per tile); each wave then multiplies its queries with every 32-row subtile:
//     QB d / 16 v_mfma_f32_32x32x16_bf16 per subtile, no barrier and no query traffic inside;
//   * a batch larger than one workgroup's queries is split into G query groups. The G
//     workgroups that stream the same corpus tiles sit on one XCD (blockIdx % 8 equal), so the
//     corpus leaves HBM once and the other G - 1 reads are hits in that XCD's L2 or in the
//     Infinity Cache (placement is a speed matter only: nothing depends on it for correctness);
//   * the epilogue never writes a score: MODE 1 compares each accumulator with its query's
//     threshold (one fma and a running maximum per accumulator; the hits are appended to
//     the query's candidate list WITH their approximate score), MODE 0 (the sample pass that
//     finds the thresholds) keeps per-(query, 32 rows) maxima.
// The bound, the sample / threshold logic and the exact f32 finish are those of trag_bf16.hip.
// After the full pass qs_prune_kernel takes, per query, the need-th largest approximate score of
// the list -- now a statistic of ALL rows, not of a sample -- and keeps only the rows within
// 2 eps of it: the exact re-score touches tens of rows per query instead of hundreds.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "trag_device.h"
#include "trag_internal.h"

// Entries the lists of one tile parity must hold together before wave 0 moves them to memory: the
// move stalls wave 0 for a round trip to memory, and the other waves at the tile's barrier; what a
// later move costs is thresholds that lag (1.25 M x 384 x 1000 queries, ms per step on one device:
// 8 entries 1.323, 16 1.299, 32 1.288; with the tighter bound 16 1.260, 48 1.232; the full pass alone,
// later: 48 entries 1.096, 64 1.112, 96 1.113, 128 1.118 -- and 1.013 with thresholds no row reaches:
// what the pass lists costs it a tenth of its time; not the drain after the stores, though: with the
// stores issued by inline assembly, which the compiler's wait counting does not see, and no drain, the
// pass took 1.072-1.081 ms against 1.072-1.075). With a ladder that never moves (more than twice the
// hits) the mirror pass takes 1.094 ms, 1.029 when the flush then drops the lists instead of writing
// them, 0.846 without any hit: three quarters of what listing costs is the listing itself -- a wave
// that has a hit runs some hundred instructions between two MFMAs, and the other seven meet it at the
// tile's barrier, nearly every tile -- and a quarter the flush. (Counting a strong row for levels
// spread down to the known one, instead of the four just below its own, changed neither the lists --
// 200-270 entries -- nor the time; nor did a raised priority, until the tile's barrier, for the wave that
// lists: 0.944 / 0.947 against 0.937 / 0.940 ms.) Moving the lists
// at the END of a tile, where wave 0 -- the older wave of its SIMD, done with its steps before the
// younger ones -- would otherwise idle at the barrier, measured no better: 1.220 against 1.213.
#ifndef QS_FLUSH_MIN
#define QS_FLUSH_MIN 48
#endif
// Staging loads by inline assembly (scalar base + 32-bit lane offset, waited for by count) instead of
// compiler-emitted loads with 64-bit vector address arithmetic per piece: 1.25 M x 384 x 1000 queries,
// sample + full pass 1.150 -> 1.124 ms from the f32 rows, 0.950 -> 0.941 ms from the mirror (same device,
// same run). -DQS_ASM_LOADS=0 for the compiler's loads.
#ifndef QS_ASM_LOADS
#define QS_ASM_LOADS 1
#endif

#ifndef QS_DMA
#define QS_DMA 1
#endif
#ifndef QS_DMA_SCHED
#define QS_DMA_SCHED 0
#endif

namespace trag {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float max_over_32_lanes(float x) {
    // lanes 0..31 and 32..63 separately: four DPP steps inside each row of 16, then the other row
    x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), 0xB1, 0xF, 0xF, false)));
    x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), 0x4E, 0xF, 0xF, false)));
    x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), 0x141, 0xF, 0xF, false)));
    x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), 0x140, 0xF, 0xF, false)));
    return fmaxf(x, __shfl_xor(x, 16));
}

// KS: k-steps of 16 (d = 16 KS). NW: waves per workgroup (8: two per SIMD, 256 registers each).
// QB: blocks of 32 queries per wave (each fragment read from LDS then feeds QB MFMAs; 2 with one
// wave per SIMD was measured slower than 1 with two: 1.96 against 1.41 ms on 1.25 M x 384 x 1000,
// the second wave hides more than the halved LDS traffic gains). RT: corpus rows per LDS tile
// (NSUB = RT / 32 subtiles). MODE 0 sample pass / 1 full pass. MIR: the corpus operand is the
// store's bf16 mirror. DP: steps a staging load is in flight. NTL: non-temporal corpus loads.
//
// The unit of work is a STEP = one 32-row subtile: d / 16 k-steps per wave, their fragments read
// through a ring of four register sets (a read is issued right behind the MFMA that frees its
// slot and is needed four k-steps later). LDS holds 2 NSUB subtile images. During step
// u the workgroup also stages: every few MFMAs a thread converts one of its 16-byte pieces of
// subtile u + NSUB (loaded DP steps ago) into that subtile's image and at once issues the load of
// the same piece DP subtiles further on -- so every load is in flight for whole steps and the
// loads in flight never drop to zero (issuing them in one burst per step left the memory system
// idle most of the time: 68 % of the wave cycles were spent waiting). One barrier per NSUB steps.
// Every tile runs the same instructions -- no variant for a stream's last tiles, whose loads are
// clamped instead -- so the loop is one basic-block chain, the staging registers need no copies
// at its back edge, and the compiler waits for a staged piece with a counted vmcnt.
// The test of a step's accumulators against the thresholds is spread over the k-steps of the NEXT
// step (see test_some).
//
// Hits (MODE 1) go to a list in LDS first (two lists, by tile parity): appending to a query's
// list in memory takes an atomic whose return the wave must wait for, and that wait drains every
// load the wave has in flight. Wave 0 empties a list into memory when it holds enough, at a time
// when nobody appends to it (the other parity's tile), and both lists at the end.
template <int KS, int NW, int QB, int RT, int MODE, bool MIR, int DP, bool NTL>
__global__ void __launch_bounds__(64 * NW, NW / 4)
qs_gemm_kernel(const float* __restrict__ corpus, const uint16_t* __restrict__ mirror,
               const __bf16* __restrict__ Qb, const float* __restrict__ rowc_g,
               uint32_t n_rows, uint32_t n_queries, uint32_t G,
               uint32_t S, uint32_t tile_step,
               SearchCtl* __restrict__ ctls, uint32_t* __restrict__ tmax,
               size_t tmax_stride, uint32_t* __restrict__ cand, uint32_t* __restrict__ cvals,
               uint32_t need, uint32_t* __restrict__ dyn_lvl, const float* __restrict__ dyn_lad,
               uint32_t* __restrict__ dyn_cnt) {
    constexpr int D = KS * 16;
    constexpr int NT = 64 * NW;
    // DMA (mirror only): the subtile's 16-byte pieces go from memory straight into LDS
    // (global_load_lds_dwordx4: a wave-instruction fills 1 KB of consecutive LDS), no staging registers,
    // no LDS write instructions. The image keeps its padded rows: a lane fetches whatever belongs at
    // the LDS position it fills (lanes that fill padding fetch the block's first bytes).
    constexpr bool DMA = MIR && (QS_DMA != 0);
    constexpr int STRIDE = D / 2 + 4;          // dwords per image row: the 4 spare dwords keep the 16-byte
                                               // fragment reads of 16 different rows on different banks
    constexpr int IMG_SLOTS = 32 * STRIDE / 4;                 // 16-byte slots of an image
    constexpr int NWI = (IMG_SLOTS + 63) / 64;                 // wave-instructions that fill one (DMA)
    constexpr int NLD = (NWI + NW - 1) / NW;                   // ... per wave, at most
    constexpr int IMG_DW = DMA ? NWI * 256 : 32 * STRIDE;      // dwords reserved per image
    constexpr int PE = MIR ? 8 : 4;            // corpus elements per 16-byte piece
    constexpr int PPR = D / PE;                // pieces per row
    constexpr int NSUB = RT / 32;              // subtiles per tile
    constexpr int NSLOT = 2 * NSUB;            // subtile images in LDS
    constexpr int NL = 32 * PPR / NT;          // pieces per thread and subtile
    constexpr int NF = 4;                      // fragment reads in flight
    constexpr int PER = KS / NL;               // MFMAs between two staged pieces
    constexpr int HITS = 64;                   // LDS hit list of one wave and tile parity: capacity
    // DP: subtiles whose loads are in flight (a load is issued DP steps before its piece is written)
    constexpr int LOOK = 1 + (NSUB - 1 + DP) / NSUB;   // the furthest tile ahead a step loads from
    static_assert(32 * PPR % NT == 0 && RT % 32 == 0 && D % 128 == 0 && KS >= NF && KS % NL == 0, "tile geometry");
    static_assert(NSUB % DP == 0 || DP == 2 * NSUB, "ring geometry");
    __shared__ __attribute__((aligned(16))) uint32_t img[NSLOT][IMG_DW];
    __shared__ __attribute__((aligned(8))) float rowc[NSLOT][32][2];   // per row: scale, shift (see below)
    __shared__ uint32_t hit_q[2][NW][MODE == 1 ? HITS : 1], hit_row[2][NW][MODE == 1 ? HITS : 1], hit_v[2][NW][MODE == 1 ? HITS : 1];
    __shared__ uint32_t hit_n[2][NW];   // entries in each list (written back by its wave at the end of every tile)
    // MODE 1, the thresholds of the workgroup's queries. They tighten during the pass (trag_device.h:
    // dyn_hit_level): wave 0 counts the hits it moves to memory for the levels of each query's ladder,
    // raises the query's level in memory when `need` rows vouch for it, and on the same trip re-reads
    // the levels of all the workgroup's queries (other workgroups raise them too) and brings this table
    // up to date; every wave takes its thresholds from the table at the start of a tile. (No barrier orders the table's writes and reads: every value a threshold ever
    // had is a valid one, a wave that reads an older one lists a few rows more.)
    constexpr int QW = 32 * NW * QB;
    __shared__ __attribute__((aligned(16))) float thr_lds[MODE == 1 ? QW : 4];
    __shared__ __attribute__((aligned(16))) uint32_t lvl_lds[MODE == 1 ? QW : 4];
    __shared__ __attribute__((aligned(16))) float lad_lds[MODE == 1 ? QW : 4][2];
    static_assert(QW == 4 * 64, "wave 0 keeps the table: four queries per lane");
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    // workgroups of one XCD: G query groups x S corpus streams
    const uint32_t b = blockIdx.x;
    const uint32_t xcd = b & 7u, slot = b >> 3;
    const uint32_t g = slot % G;
    const uint32_t strm = (slot / G) * 8u + xcd, n_streams = S * 8u;
    // tiles of the pass: every tile_step-th of the corpus; unit e of the pass is corpus tile e * tile_step.
    // This stream takes the units strm, strm + n_streams, ...: n_mine of them.
    const uint32_t n_tiles = (n_rows + RT - 1) / RT;
    const uint32_t n_units = (n_tiles + tile_step - 1) / tile_step;
    if (strm >= n_units) return;
    const uint32_t n_mine = (n_units - strm + n_streams - 1) / n_streams;
    // The G workgroups of a stream walk its units in the same order. (With QS_ROTATE: in rotated order
    // inside groups of G -- group c: c G + (j + g) % G for j = 0 .. G - 1 -- so that at any time they ask
    // memory for G different tiles; a last, partial group is walked in plain order.)
    // (Measured, 1.25 M x 384 x 1000 queries, G = 4: walked in rotated order the f32 rows cost 3.61 GB of
    // FETCH_SIZE for 1.92 GB of corpus and the mirror 1.115 GB for 0.96; in plain order -- the G workgroups
    // ask for the same tile at about the same time, one miss and G - 1 hits -- 2.63 GB and 1.014 GB, and the
    // step is 2 % / 1 % shorter. -DQS_ROTATE brings the rotation back.)
#ifdef QS_ROTATE
    const uint32_t n_rot = (n_mine / G) * G;
#else
    const uint32_t n_rot = 0u;
#endif
    // the units in this workgroup's order, one after the other (no division in the loop); past the
    // last one: NO_UNIT
    constexpr uint32_t NO_UNIT = 0xFFFFFFFFu;
    uint32_t gen_n = 0, gen_c = 0, gen_j = 0;
    auto next_unit = [&]() __attribute__((always_inline)) -> uint32_t {
        uint32_t m = gen_n;
        if (gen_n < n_rot) {
            uint32_t jj = gen_j + g;
            if (jj >= G) jj -= G;
            m = gen_c + jj;
        }
        const bool valid = gen_n < n_mine;
        gen_n += 1;
        gen_j += 1;
        if (gen_j == G) { gen_j = 0; gen_c += G; }
        return valid ? m * n_streams + strm : NO_UNIT;
    };
    const uint32_t rows_per_unit = tile_step * RT;
    const uint32_t q0 = g * (32u * NW * QB) + (uint32_t)wid * (32u * QB);   // this wave's queries: q0 .. q0 + 32 QB - 1

    // Wait for every vector-memory operation of this wave (s_waitcnt vmcnt(0) only). Loads return in
    // order, so the compiler can wait for "all but the newest N"; stores and atomics return out of
    // order with them, and one of those possibly in flight -- even on a path that is almost never
    // taken -- turns every later wait in the loop into a wait for ALL loads, the newest included.
    // Each rare path that writes to memory therefore ends with this, and the staging loads of the
    // common path keep their counted waits.
    auto drain_vmem = [&]() __attribute__((always_inline)) { __builtin_amdgcn_s_waitcnt(0x0F70); };
    // the queries' fragments, for the whole K extent, for the life of the workgroup
    bf16x8 qa[QB][KS];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const __bf16* qp = Qb + (size_t)(q0 + 32u * (uint32_t)qb + (uint32_t)r) * D + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            qa[qb][ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(qp + ks * 16));
    }
#ifndef QS_PIN_KS0
#define QS_PIN_KS0 64
#endif
#ifdef QS_PIN_AGPR
    // One wave per SIMD (512 registers): the query fragments live in the accumulation half of the
    // register file for the whole kernel and the MFMAs read them there (their A / B operands may be
    // AGPRs on gfx950). Left to itself the compiler parks them there too -- as spills, and copies four
    // registers back in front of every MFMA.
    if (NW == 4) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (qb == 0 && ks >= QS_PIN_KS0) continue;
                u32x4 t = __builtin_bit_cast(u32x4, qa[qb][ks]);
                asm volatile("" : "=a"(t) : "0"(t));
                qa[qb][ks] = __builtin_bit_cast(bf16x8, t);
            }
    }
#endif
    // accumulator register rr of block qb of lane (r, h) belongs to query q0 + 32 qb + qrow(rr) and corpus
    // row r of the subtile; e = 16 qb + rr numbers a lane's accumulators
    auto qrow = [&](int rr) __attribute__((always_inline)) { return (uint32_t)((rr & 3) + 8 * (rr >> 2) + 4 * h); };
    constexpr int NA = 16 * QB;
    float th[NA];
    const uint32_t qg0 = g * (uint32_t)QW;   // first query of the workgroup
    auto load_thresholds = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int blk = 0; blk < QB; ++blk)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const v4f x = *reinterpret_cast<const v4f*>(&thr_lds[wid * 32 * QB + 32 * blk + 8 * g4 + 4 * h]);
#pragma unroll
                for (int e = 0; e < 4; ++e) th[16 * blk + 4 * g4 + e] = x[e];
            }
    };
    if (MODE == 1) {
        if (tid < QW) {
            // (the ladder arrays are padded to whole query groups: +inf behind the batch)
            const uint32_t q = qg0 + (uint32_t)tid;
            const uint32_t lv = dyn_lvl[q];
            float t0 = dyn_lad[2 * q];
            const float step = dyn_lad[2 * q + 1];
#ifdef QS_EXP_NOHIT
            t0 = INFINITY;
#endif
            // A query outside the bound's range (threshold -inf) would list every row: its list is
            // marked as overflowed once, here, and it takes no further part (the final stage reports
            // the overflow and the host repeats the query with the exact scan).
            if (t0 == -INFINITY) {
                if (strm == 0u) atomicMax(&ctls[q].cand_count, CAND_CAP + 1u);
                t0 = INFINITY;
            }
            lvl_lds[tid] = lv;
            lad_lds[tid][0] = t0;
            lad_lds[tid][1] = step;
            thr_lds[tid] = dyn_level_value(lv, t0, step);
        }
        if (tid < 2 * NW) hit_n[tid / NW][tid % NW] = 0u;
        drain_vmem();
        __syncthreads();
        load_thresholds();
    }

    // staging: the 32 rows of a subtile are one contiguous block of memory, cut into 16-byte pieces;
    // piece i * NT + tid is this thread's i-th (consecutive lanes, consecutive pieces: coalesced)
    uint32_t loff[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int p = i * NT + tid, row = p / PPR, c = p % PPR;
        loff[i] = (uint32_t)(row * STRIDE + c * (MIR ? 4 : 2));
    }
    u32x4 pre[DP][NL];
    // every thread also carries the (scale, shift) pair of row (tid & 31) of the staged subtile (32
    // threads would do; all of them doing the same keeps the step free of branches, which is what
    // lets the compiler count the loads in flight instead of draining them)
    float2 pre_rc[DP];
#pragma unroll
    for (int k = 0; k < DP; ++k) pre_rc[k] = make_float2(0.0f, -INFINITY);
    // load piece i of the subtile that starts at row0 into slot k of the staging registers. The
    // subtile is one contiguous block of memory: its address is a wave-uniform base (scalar registers)
    // plus a per-thread byte offset fixed for the kernel's life, so a piece costs no address
    // arithmetic in the vector ALU. Nothing is clamped: the store's row buffers (f32 rows, mirror)
    // hold a multiple of 64 rows and rowc_g is written up to that multiple, so the rows of a partial
    // last tile are readable memory; what they hold does not matter, their scale / shift say
    // "takes no part". (A unit past the stream's last is read from tile 0 and never used.) No branch:
    // the same instructions run for every tile, which keeps the loop a single body.
    const char* const src = MIR ? reinterpret_cast<const char*>(mirror) : reinterpret_cast<const char*>(corpus);
    constexpr size_t ROW_BYTES = (size_t)D * (MIR ? 2 : 4);
    const uint32_t voff = (uint32_t)tid * 16u;
    const uint32_t rc_off = (uint32_t)(tid & 31) * 8u;
    auto issue_piece = [&](int k, int i, uint32_t row0) __attribute__((always_inline)) {
#ifdef QS_EXP_NOGLOAD
        return;   // (ablation: LDS is written from registers that are never refreshed)
#endif
        const char* base = src + (size_t)row0 * ROW_BYTES + (size_t)(i * NT) * 16u;   // (uniform)
#if QS_ASM_LOADS && defined(__HIP_DEVICE_COMPILE__)
        // scalar base + 32-bit lane offset: no 64-bit vector address arithmetic per piece. Not known to the
        // compiler as loads: write_piece waits for them by count.
        if (NTL) asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(pre[k][i]) : "v"(voff), "s"(base) : "memory");
        else asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(pre[k][i]) : "v"(voff), "s"(base) : "memory");
        if (i == 0) {
            const char* rb = reinterpret_cast<const char*>(rowc_g) + (size_t)row0 * 8u;
            asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(pre_rc[k]) : "v"(rc_off), "s"(rb) : "memory");
        }
#else
        const u32x4* p = reinterpret_cast<const u32x4*>(base + voff);
        pre[k][i] = NTL ? __builtin_nontemporal_load(p) : *p;
        if (i == 0)
            pre_rc[k] = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(rowc_g) + (size_t)row0 * 8u + rc_off);
#endif
    };
    // (scale, shift): see row_scale_shift (trag_device.h)
    auto write_piece = [&](int k, int i, int sl, uint32_t) __attribute__((always_inline)) {
#ifdef QS_EXP_NOLDSWRITE
        // (ablation: the loads are issued and waited for, nothing is written to LDS)
        asm volatile("" ::"v"(pre[k][i]));
        if (i == 0) asm volatile("" ::"v"(pre_rc[k].x), "v"(pre_rc[k].y));
        return;
#endif
#if QS_ASM_LOADS && defined(__HIP_DEVICE_COMPILE__)
        // a piece was loaded DP steps ago; since then the wave has issued DP (NL + 1) - 1 younger loads
        // (pieces and the rows' constants, in that order behind piece 0), and loads return in order
        if (i == 0) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(pre[k][i]), "+v"(pre_rc[k]) : "n"(DP * (NL + 1) - 2) : "memory");
        else asm volatile("s_waitcnt vmcnt(%1)" : "+v"(pre[k][i]) : "n"(DP * (NL + 1) - 1) : "memory");
#endif
        if (MIR) {
            *reinterpret_cast<u32x4*>(&img[sl][loff[i]]) = pre[k][i];
        } else {
            const bf16x4 hv = __builtin_convertvector(__builtin_bit_cast(v4f, pre[k][i]), bf16x4);
            *reinterpret_cast<u32x2*>(&img[sl][loff[i]]) = __builtin_bit_cast(u32x2, hv);
        }
        if (i == 0) *reinterpret_cast<float2*>(&rowc[sl][tid & 31][0]) = pre_rc[k];   // (16 threads store the same pair: harmless)
    };

    // DMA form: 16-byte slot i * NT + tid of a subtile's image comes from byte goff[i] of the subtile's
    // block of memory; the 64 slots of one wave-instruction are 1 KB of consecutive LDS, whose start
    // goes into M0. The rows' (scale, shift) pairs travel the same way (256 bytes, every wave asks for
    // them: the same bytes to the same place).
    uint32_t goff[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int p = i * NT + tid, row = p / (PPR + 1), c = p % (PPR + 1);
        goff[i] = (row < 32 && c < PPR) ? (uint32_t)(row * (int)ROW_BYTES + c * 16) : 0u;
    }
    typedef __attribute__((address_space(3))) uint32_t* lds_u32;
    const int wid_s = __builtin_amdgcn_readfirstlane(wid);
    [[maybe_unused]] const uint32_t img_lds = (uint32_t)(uintptr_t)(lds_u32)(&img[0][0]) + (uint32_t)wid_s * 1024u;
    [[maybe_unused]] const uint32_t rowc_lds = (uint32_t)(uintptr_t)(lds_u32)(reinterpret_cast<uint32_t*>(&rowc[0][0][0]));
    [[maybe_unused]] const uint32_t lane4 = (uint32_t)lane * 4u;
    constexpr int NDMA = NSUB * (NLD + 1);     // transfers per wave and tile
    // transfer j of the tile that starts at row `trow` into the images of half `hf`
    auto dma_piece = [&](int j, uint32_t trow, int hf) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
        const int sub = j / (NLD + 1), i = j % (NLD + 1);
        const uint32_t row0 = trow + 32u * (uint32_t)sub;
        if (i < NLD) {
            if ((i + 1) * NW > NWI && i * NW + wid_s >= NWI) return;   // (the last round is a partial one)
            const char* base = src + (size_t)row0 * ROW_BYTES;   // (uniform)
            const uint32_t m = img_lds + (uint32_t)((hf * NSUB + sub) * IMG_DW * 4 + i * NT * 16);
            if (NTL) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt" :: "s"(m), "v"(goff[i]), "s"(base) : "memory");
            else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m), "v"(goff[i]), "s"(base) : "memory");
        } else {
            const char* rb = reinterpret_cast<const char*>(rowc_g) + (size_t)row0 * 8u;
            const uint32_t m = rowc_lds + (uint32_t)((hf * NSUB + sub) * 32 * 8);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" :: "s"(m), "v"(lane4), "s"(rb) : "memory");
        }
#endif
    };

    bf16x8 fr[NF];     // the fragment ring
    auto frag = [&](int sl, int ks) __attribute__((always_inline)) {
        return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&img[sl][r * STRIDE + 4 * h + ks * 8]));
    };
    // Hits go to lists in LDS first, one per wave and tile parity, so that listing needs no atomic
    // and no wait: `hits_here` is the wave's own count for the current tile's parity (wave-uniform,
    // read from hit_n at the start of a tile and written back at its end). Wave 0 empties the lists of
    // the OTHER parity into the queries' lists in memory, at the start of a tile, when one of them is
    // a quarter full: nobody appends to them during that tile, and only wave 0's loads are drained by the
    // atomics' returns. A wave whose own list is full (a burst of hits: a row that scores high for most
    // queries at once) empties it into memory itself before it goes on.
    uint32_t hits_here = 0u;
    auto flush_own = [&](int par) __attribute__((always_inline)) {
        if ((uint32_t)lane < hits_here) {
            const uint32_t q = hit_q[par][wid][lane];
            const uint32_t gp = atomicAdd(&ctls[q].cand_count, 1u);
            if (gp < CAND_CAP) {
                cand[(size_t)q * CAND_CAP + gp] = hit_row[par][wid][lane];
                cvals[(size_t)q * CAND_CAP + gp] = hit_v[par][wid][lane];
            }
        }
        drain_vmem();
        hits_here = 0u;
    };
    auto flush = [&](int par) __attribute__((always_inline)) {
        // The NW lists are taken as one sequence, 64 entries a round (a round is one round trip to
        // memory -- wave 0 is stalled for that long, and at the tile's barrier so is everybody else):
        // all the atomics first -- the list cursor and the counters of the ladder levels the row
        // vouches for above the level the table knows -- then the stores that need their results.
        int ln = lane;
        asm volatile("" : "+v"(ln));   // (see run_tile)
        const uint32_t fill = ln < NW ? min(hit_n[par][ln], (uint32_t)HITS) : 0u;
        uint32_t start[NW + 1];
        start[0] = 0u;
#pragma unroll
        for (int w = 0; w < NW; ++w) start[w + 1] = start[w] + (uint32_t)__builtin_amdgcn_readlane((int)fill, w);
        const uint32_t total = start[NW];
        // (the queries' levels, as other workgroups have raised them meanwhile: the read travels with
        // the atomics below)
        const u32x4 lv = *reinterpret_cast<const u32x4*>(&dyn_lvl[qg0 + 4u * (uint32_t)ln]);
        const uint32_t* hq = &hit_q[par][0][0];
        const uint32_t* hr = &hit_row[par][0][0];
        const uint32_t* hv = &hit_v[par][0][0];
        for (uint32_t base = 0; base < total; base += 64u) {
            const uint32_t idx = base + (uint32_t)ln;
            const bool valid = idx < total;
            uint32_t w = 0u, s0 = 0u;
#pragma unroll
            for (int ww = 1; ww < NW; ++ww) {
                const bool ge = idx >= start[ww];
                w = ge ? (uint32_t)ww : w;
                s0 = ge ? start[ww] : s0;
            }
            const uint32_t li = valid ? w * (uint32_t)HITS + (idx - s0) : 0u;
            const uint32_t q = hq[li], row = hr[li], vbits = hv[li];
            const uint32_t ql = valid ? q - qg0 : 0u;
            const int jh = valid ? dyn_hit_level(__uint_as_float(vbits), lad_lds[ql][0], lad_lds[ql][1], (int)DYN_LEVELS, (int)DYN_SUB) : -1;
            const int jlo = max((int)lvl_lds[ql], jh - (int)DYN_MAXINC);
            uint32_t old[DYN_MAXINC];
#pragma unroll
            for (int i = 0; i < (int)DYN_MAXINC; ++i)
                old[i] = (valid && jh - i > jlo) ? atomicAdd(&dyn_cnt[(size_t)q * DYN_LEVELS + (uint32_t)(jh - i)], 1u) : 0xFFFFFFF0u;
            const uint32_t pos = valid ? atomicAdd(&ctls[q].cand_count, 1u) : CAND_CAP;
            if (pos < CAND_CAP) {
                cand[(size_t)q * CAND_CAP + pos] = row;
                cvals[(size_t)q * CAND_CAP + pos] = vbits;
            }
#pragma unroll
            for (int i = 0; i < (int)DYN_MAXINC; ++i)
                if (old[i] + 1u == need) atomicMax(&dyn_lvl[q], (uint32_t)(jh - i));
        }
        {
            const int q4 = 4 * ln;
            v4f t;
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i] = dyn_level_value(lv[i], lad_lds[q4 + i][0], lad_lds[q4 + i][1]);
            *reinterpret_cast<v4f*>(&thr_lds[q4]) = t;
            *reinterpret_cast<u32x4*>(&lvl_lds[q4]) = lv;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (ln < NW) hit_n[par][ln] = 0u;
        drain_vmem();
    };
    static_assert(HITS == 64, "a wave's hits of one round always fit an empty list");
    // MODE 0: the largest value of each (query, subtile), as an ordered key
    auto epilogue = [&](const v16f (&acc)[QB], int sl, uint32_t idx) __attribute__((always_inline)) {
        const float2 rc = *reinterpret_cast<const float2*>(&rowc[sl][r][0]);
        float sc = rc.x, sh = rc.y;
        {
            if (sh == INFINITY) { sc = 0.0f; sh = -INFINITY; }
            // (the query index passes through an opaque move: the output addresses are then
            // formed here, not hoisted out of the step loop into registers for its whole length)
            uint32_t qb = q0 + 4u * (uint32_t)h;
            asm volatile("" : "+v"(qb));
#pragma unroll
            for (int e = 0; e < NA; ++e) {
                const int rr = e & 15;
                const float best = max_over_32_lanes(__builtin_fmaf(acc[e >> 4][rr], sc, sh));
                const uint32_t q = qb + (uint32_t)(32 * (e >> 4) + (rr & 3) + 8 * (rr >> 2));
                if (r == 0 && q < n_queries)
                    tmax[q * tmax_stride + idx] = (best > -INFINITY && best < INFINITY) ? score_key(best) : 0u;
            }
        }
    };
    // MODE 1. Whether lane (r, h) has a hit among its sixteen accumulators: the value of accumulator
    // rr is v = fma(acc, sc, sh) and a hit is v >= th[rr]; sh is the same for all sixteen (one corpus
    // row per lane), so the test is  max over rr of fma(acc[rr], sc, -th[rr])  + sh >= 0: two
    // operations per accumulator, whatever the metric. (For the Euclidean metric this sum is rounded
    // in another order than v - th: a difference of a few units in the last place, against a
    // threshold that keeps 1 % of the bound as slack.) These operations do not wait for a step's end:
    // the accumulators of step u are tested, a few at a time, BETWEEN the MFMAs of step u + 1, where
    // the vector ALU is idle anyway; done after the step they would stall the matrix pipe of both
    // waves of the SIMD, which reach their step ends together.
    // The running maximum is kept per group of four accumulators (NG maxima): the same number of
    // operations, and the continuation below then looks only at the groups that can hold a hit.
    constexpr int NG = NA / 4;
    auto test_some = [&](const v16f (&acc)[QB], float sc, int first, int count, float (&m)[NG]) __attribute__((always_inline)) {
#pragma unroll
        for (int e = first; e < first + count && e < NA; ++e)
            m[e >> 2] = fmaxf(m[e >> 2], __builtin_fmaf(acc[e >> 4][e & 15], sc, -th[e]));   // (fmaxf drops a NaN operand)
    };
    // ... and the continuation (a few tenths of the steps at 10 wanted rows per query and a million
    // corpus rows): list the hits. A NaN counts as a hit; a row that takes no part (scale 0 and
    // shift -inf) has none. The branch is taken by the whole wave or not at all, the hit bits of a
    // lane are formed without branches, and the lanes that have one write it at consecutive
    // positions of the wave's own list: counting by ballot, no atomic, nothing to wait for.
    uint32_t qmask = 0u;   // bit e: accumulator e belongs to a query of the batch
#pragma unroll
    for (int e = 0; e < NA; ++e) {
        const uint32_t q = q0 + 32u * (uint32_t)(e >> 4) + qrow(e & 15);
        qmask |= q < n_queries ? (1u << e) : 0u;
    }
    static_assert(NA <= 32, "hit bits of a lane fit one word");
    auto list_hits = [&](const v16f (&acc)[QB], float sc, float sh, const float (&m)[NG], uint32_t row0, int par) __attribute__((always_inline)) {
        float mall = m[0];
#pragma unroll
        for (int g = 1; g < NG; ++g) mall = fmaxf(mall, m[g]);
        if (__builtin_amdgcn_ballot_w64(!(mall + sh < 0.0f)) != 0ull) {
            const bool live = sh != -INFINITY || sc != 0.0f;
            uint32_t w = 0u;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (__builtin_amdgcn_ballot_w64(!(m[g] + sh < 0.0f)) != 0ull) {   // (typically one group of the wave)
#pragma unroll
                    for (int e = 4 * g; e < 4 * g + 4; ++e)
                        w |= !(__builtin_fmaf(acc[e >> 4][e & 15], sc, -th[e]) + sh < 0.0f) ? (1u << e) : 0u;
                }
            }
            w = live ? (w & qmask) : 0u;
            const uint32_t row = row0 + (uint32_t)r;
            uint32_t qb = q0 + 4u * (uint32_t)h;
            unsigned long long any = __builtin_amdgcn_ballot_w64(w != 0u);
            while (any != 0ull) {
                const bool act = w != 0u;
                const uint32_t bit = act ? (uint32_t)__builtin_ctz(w) : 0xFFFFFFFFu;
                w &= w - 1u;
                float a = 0.0f;   // accumulator `bit` (selects: the index differs by lane; only the groups in play)
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (__builtin_amdgcn_ballot_w64((bit >> 2) == (uint32_t)g) != 0ull) {
#pragma unroll
                        for (int e = 4 * g; e < 4 * g + 4; ++e) a = bit == (uint32_t)e ? acc[e >> 4][e & 15] : a;
                    }
                }
                const float v = __builtin_fmaf(a, sc, sh);
                const uint32_t add = (uint32_t)__builtin_popcountll(any);   // <= 64 = the list's capacity
                if (hits_here + add > (uint32_t)HITS) flush_own(par);
                const uint32_t pos = hits_here + __builtin_amdgcn_mbcnt_hi((uint32_t)(any >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)any, 0u));
                if (act) {
                    const uint32_t rr = bit & 15u;
                    hit_q[par][wid][pos] = qb + 32u * (bit >> 4) + (rr & 3u) + 8u * (rr >> 2);
                    hit_row[par][wid][pos] = row;
                    hit_v[par][wid][pos] = __float_as_uint(v);
                }
                hits_here += add;
                any = __builtin_amdgcn_ballot_w64(w != 0u);
            }
        }
    };

    // prologue: the first NSUB subtiles into their images, the next DP into the staging registers.
    // A unit past the stream's last is loaded like any other (from row 0: its data is never used).
    uint32_t unit[LOOK + 1];   // tiles n .. n + LOOK
#pragma unroll
    for (int t = 0; t <= LOOK; ++t) {
        const uint32_t u = next_unit();
        unit[t] = u == NO_UNIT ? 0u : u;
    }
    if (DMA) {
#pragma unroll
        for (int j = 0; j < NDMA; ++j) dma_piece(j, unit[0] * rows_per_unit, 0);
        drain_vmem();
    } else {
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            const uint32_t row0 = unit[0] * rows_per_unit + 32u * (uint32_t)sub;
#pragma unroll
            for (int i = 0; i < NL; ++i) issue_piece(0, i, row0);
            drain_vmem();
#pragma unroll
            for (int i = 0; i < NL; ++i) write_piece(0, i, sub, row0);
        }
#pragma unroll
        for (int w = NSUB; w < NSUB + DP; ++w) {   // subtile w of the stream: tile w / NSUB, ring slot w % DP
#pragma unroll
            for (int i = 0; i < NL; ++i)
                issue_piece(w % DP, i, unit[w / NSUB] * rows_per_unit + 32u * (uint32_t)(w % NSUB));
        }
    }
    __syncthreads();

    // One tile = NSUB steps + the barrier. Step `sub` of tile n is step u = n NSUB + sub of the stream:
    // the workgroup writes subtile u + NSUB (subtile `sub` of tile n + 1) from ring slot
    // (u + NSUB) % DP and loads subtile u + NSUB + DP into the same slot. ODD: parity of tile n + 1
    // (only matters when the ring spans two tiles).
    // MODE 1: the step before this one, whose accumulators are still to be tested (at the start: none:
    // a row that takes no part)
    v16f accp[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int e = 0; e < 16; ++e) accp[qb][e] = 0.0f;
    float scp = 0.0f, shp = -INFINITY, mp[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) mp[g] = -INFINITY;
    uint32_t row0p = 0u;
    constexpr int NCH = KS < 16 ? KS : 16;       // k-steps of a step that are followed by a piece of the test
    constexpr int CH = (NA + NCH - 1) / NCH;     // accumulators per piece
#ifdef QS_STAMP
    unsigned long long st_work = 0, st_bar = 0, st_t0 = __builtin_amdgcn_s_memtime(), st_all0 = st_t0;
#endif
    auto run_tile = [&](auto odd_tag, uint32_t n) __attribute__((always_inline)) {
        constexpr int ODD = decltype(odd_tag)::value ? 1 : 0;
        const int half = (int)(n & 1u);   // which half of the images this tile's subtiles are in; its hit list
        if (MODE == 1) {
            hits_here = hit_n[half][wid];
            if (wid == 0) {
                // the other parity's lists -> memory, when they hold enough (nobody appends to them during
                // this tile)
                // (the lane index passes through an opaque move wherever wave 0 forms an address that is the
                // same for every tile: hoisted out of the loop, such an address is one more value live across
                // the whole step loop, which has no register to spare -- it would be spilled, and its reload is
                // a vector-memory load whose wait drains every staging load of the wave)
                int ln = lane;
                asm volatile("" : "+v"(ln));
                const uint32_t fill = ln < NW ? hit_n[half ^ 1][ln] : 0u;
                uint32_t total = 0u;
#pragma unroll
                for (int w = 0; w < NW; ++w) total += (uint32_t)__builtin_amdgcn_readlane((int)fill, w);
#ifdef QS_FLUSH_EARLY
                // the first tiles of a stream run on the sample's thresholds, the loosest of the pass: what
                // they list reaches the ladder's counters at once
                if (total >= (n < (uint32_t)QS_FLUSH_EARLY ? 4u : (uint32_t)QS_FLUSH_MIN)) flush(half ^ 1);
#else
                if (total >= (uint32_t)QS_FLUSH_MIN) flush(half ^ 1);
#endif
            }
            load_thresholds();
        }
#pragma unroll
        for (int j = 0; j < NF; ++j) fr[j] = frag(half * NSUB, j);
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            const int sl = half * NSUB + sub;            // image of this step
            const int wsl = (half ^ 1) * NSUB + sub;     // image written during this step
            const int ring = (ODD * NSUB + sub) % DP;    // (a constant after unrolling)
            const int it = 1 + (sub + DP) / NSUB, isub = (sub + DP) % NSUB;   // tile ahead / subtile of the load
            const uint32_t row0 = unit[0] * rows_per_unit + 32u * (uint32_t)sub;
            const uint32_t row0_write = unit[1] * rows_per_unit + 32u * (uint32_t)sub;
#ifdef QS_EXP_SAMETILE
            const uint32_t row0_issue = 32u * (uint32_t)isub + 0u * unit[it];   // (every load hits L2: latency ablation)
#else
            const uint32_t row0_issue = unit[it] * rows_per_unit + 32u * (uint32_t)isub;
#endif
#ifdef QS_PRIO
            // the two waves of a SIMD take turns at being preferred by the issue arbiter (which, left
            // alone, always prefers the older one: waves 0..3). (Measured without effect, as was one
            // s_setprio 1 for waves 4..7 before the tile loop: 1.196 / 1.197 ms per step without, 1.202 /
            // 1.191 with, 1.25 M x 384 x 1000 queries.)
            if (NW == 8) {
                if ((wid >= 4) == ((sub & 1) != 0)) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
            }
#endif
            v16f acc[QB];
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[qb][e] = 0.0f;
            auto k_step = [&](int ks) __attribute__((always_inline)) {
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
                    acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[qb][ks], fr[ks % NF], acc[qb], 0, 0, 0);
                if (ks + NF < KS) fr[ks % NF] = frag(sl, ks + NF);
#ifndef QS_EXP_NOSTAGE
                if (DMA) {
                    // the whole next tile is asked for during this tile's first step, a transfer every few
                    // k-steps: each has the rest of the tile to land (waited for in front of the barrier)
#if QS_DMA_SCHED == 1
                    // staggered: even waves ask at even k-steps, odd waves at odd ones (the waves of a workgroup run
                    // in step behind the barrier: asking at the same k-step, all eight queue up at the one
                    // address unit of the CU)
                    {
                        const int u = sub * KS + ks;
                        if (u / 2 < NDMA && (u & 1) == (wid_s & 1)) dma_piece(u / 2, unit[1] * rows_per_unit, half ^ 1);
                    }
#elif QS_DMA_SCHED == 2
                    // all at once, at the top of the tile
                    if (sub == 0 && ks == 0) {
#pragma unroll
                        for (int j = 0; j < NDMA; ++j) dma_piece(j, unit[1] * rows_per_unit, half ^ 1);
                    }
#elif QS_DMA_SCHED == 3
                    // four classes of waves, one k-step apart
                    {
                        const int u = sub * KS + ks;
                        if (u / 4 < NDMA && (u & 3) == (wid_s & 3)) dma_piece(u / 4, unit[1] * rows_per_unit, half ^ 1);
                    }
#else
                    if (sub == 0) {
#pragma unroll
                        for (int j = 0; j < NDMA; ++j)
                            if (ks == (j * KS) / NDMA) dma_piece(j, unit[1] * rows_per_unit, half ^ 1);
                    }
#endif
                } else if (ks % PER == PER - 1) {
                    write_piece(ring, ks / PER, wsl, row0_write);
                    issue_piece(ring, ks / PER, row0_issue);
                }
#endif
            };
            // (two loops, so that the listing of hits is not part of a loop body: with it inside, the
            // body times its trip count passes the size up to which the compiler unrolls fully, and the
            // fragment registers end up indexed at run time, in memory)
#pragma unroll
            for (int ks = 0; ks < (MODE == 1 ? NCH : 0); ++ks) {
                k_step(ks);
#ifndef QS_EXP_NOTEST
                test_some(accp, scp, ks * CH, CH, mp);
#endif
                // (pins this piece of the test to this k-step: without it the compiler sinks all the
                // pieces to the point where the running maximum is used and runs them in one burst
                // between two MFMAs)
                asm volatile("" :: "v"(mp[(ks * CH) >> 2]));
                __builtin_amdgcn_sched_barrier(0);
            }
#ifndef QS_EXP_NOTEST
            if (MODE == 1) list_hits(accp, scp, shp, mp, row0p, half);
#elif defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) asm volatile("" :: "v"(accp[qb]));   // (keeps the products alive)
#endif
#pragma unroll
            for (int ks = (MODE == 1 ? NCH : 0); ks < KS; ++ks) {
                k_step(ks);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (sub + 1 < NSUB) {
#pragma unroll
                for (int j = 0; j < NF; ++j) fr[(KS + j) % NF] = frag(sl + 1, j);
            }
            if (MODE == 0) {
                epilogue(acc, sl, unit[0] * (uint32_t)NSUB + (uint32_t)sub);   // (index among the sampled subtiles)
            } else {
                const float2 rc = *reinterpret_cast<const float2*>(&rowc[sl][r][0]);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) accp[qb] = acc[qb];
                scp = rc.x;
                shp = rc.y;
#pragma unroll
                for (int g = 0; g < NG; ++g) mp[g] = -INFINITY;
                row0p = row0;
            }
        }
        if (MODE == 1 && lane == 0) hit_n[half][wid] = min(hits_here, (uint32_t)HITS);
#ifdef QS_STAMP
        {
            const unsigned long long t1 = __builtin_amdgcn_s_memtime();
            st_work += t1 - st_t0;
            __syncthreads();
            const unsigned long long t2 = __builtin_amdgcn_s_memtime();
            st_bar += t2 - t1;
            st_t0 = t2;
        }
#elif defined(QS_EXP_NOBAR)
#else
        if (DMA) drain_vmem();   // (the next tile's transfers have landed)
        __syncthreads();
#endif
    };
    constexpr bool BY_PARITY = DP > NSUB;
    for (uint32_t n = 0; n < n_mine; ++n) {
        if (BY_PARITY && ((n + 1u) & 1u)) run_tile(std::integral_constant<bool, true>{}, n);
        else run_tile(std::integral_constant<bool, false>{}, n);
#pragma unroll
        for (int t = 0; t < LOOK; ++t) unit[t] = unit[t + 1];
        const uint32_t u = next_unit();
        unit[LOOK] = u == NO_UNIT ? 0u : u;
    }
#ifdef QS_STAMP
    if (MODE == 1 && lane == 0 && (blockIdx.x % 67u) == 5u)
        printf("blk %u wave %d tiles %u work %llu bar %llu total %llu\n", blockIdx.x, wid, n_mine, st_work, st_bar,
               (unsigned long long)__builtin_amdgcn_s_memtime() - st_all0);
#endif
    if (MODE == 1) {
        // the last step's accumulators; then, when every wave has listed its hits, both lists
        hits_here = hit_n[0][wid];
        test_some(accp, scp, 0, NA, mp);
        list_hits(accp, scp, shp, mp, row0p, 0);
        if (lane == 0) hit_n[0][wid] = min(hits_here, (uint32_t)HITS);
        __syncthreads();
        if (wid == 0) {
            flush(0);
            flush(1);
        }
    }
}

// Per query, after the full pass: the candidate list holds every row whose approximate score v
// reaches the sample's threshold, with v. A = the need-th largest finite v of the list: `need`
// rows reach A approximately, so the exact need-th best is >= A - eps and every row of the exact
// top `need` has v >= A - 2 eps. Keep those rows (and the rows whose v is +inf or NaN: rows
// outside the bound's range, always scored exactly), drop the rest, and record the lowered
// threshold for the final check. A list that overflowed is left alone (the final stage reports it).
// The full pass may have raised the query's threshold along its ladder (trag_device.h): the rows
// left out of the list are then only known to lie below that raised value, and the final check
// must be made against it.
// The list is ranked by counting: a thread keeps its E entries in registers and counts, over all
// the list's keys in LDS, the entries that sort before each (larger value first, earlier entry
// first among equals) -- two barriers where a sorting network for 256 entries takes 36, and 32 KB of
// LDS where the network's keys and the rows took 48 KB, three workgroups to a CU.
// The survivors -- a prefix of the sorted order -- land where the sort would have put them.
// (256 threads: the lists are a few hundred entries long, and four times as many workgroups fit a CU
// as with 1024 -- what the stage costs is the latency of its loads and barriers, not its work.)
constexpr int PRUNE_THREADS = 256;

template <int METRIC, int E>
__device__ __forceinline__ void prune_list(SearchCtl* __restrict__ ctl, uint32_t* __restrict__ cq,
                                           const uint32_t* __restrict__ vq, float* __restrict__ thr_y, uint32_t d,
                                           uint32_t C, uint32_t need, float na, float na_sq, float tv_cur,
                                           unsigned long long* keys_sh, uint32_t* shared3) {
    const int tid = threadIdx.x;
    // an entry's sort key: its value's ordered key, then earlier entries first -- unique, so one
    // 64-bit comparison per pair ranks the list (the counting is the stage's arithmetic: every
    // thread against every entry)
    unsigned long long key[E];
    uint32_t row[E], above[E];
    uint32_t n_inf = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const uint32_t i = (uint32_t)tid + (uint32_t)e * PRUNE_THREADS;
        key[e] = 0ull;
        row[e] = 0u;
        above[e] = 0u;
        if (i < C) {
            const float v = __uint_as_float(vq[i]);
            row[e] = cq[i];
            // +inf and NaN sort first: they are kept whatever the threshold
            const uint32_t k32 = (v != v || v == INFINITY) ? 0xFFFFFFFFu : score_key(v);
            n_inf += k32 == 0xFFFFFFFFu;
            key[e] = ((unsigned long long)k32 << 32) | (unsigned long long)(0xFFFFFFFFu - i);
            keys_sh[i] = key[e];
        }
    }
    if (n_inf) atomicAdd(&shared3[0], n_inf);
    __syncthreads();
    const uint32_t F = shared3[0];
    if (F + need > C) return;   // fewer than `need` rows with a usable score: keep the whole list
    for (uint32_t j = 0; j < C; ++j) {
        const unsigned long long o = keys_sh[j];
#pragma unroll
        for (int e = 0; e < E; ++e) above[e] += o > key[e] ? 1u : 0u;
    }
#pragma unroll
    for (int e = 0; e < E; ++e)
        if ((uint32_t)tid + (uint32_t)e * PRUNE_THREADS < C && above[e] == F + need - 1u) shared3[2] = (uint32_t)(key[e] >> 32);
    __syncthreads();
    // A = the need-th largest usable value; the lowered threshold, in score space (t, for the final
    // check) and in v space (tv)
    const float A = key_score(shared3[2]);
    const float vmax = sqrtf(ctl->vmax_sq);
    const float eps = bf16_eps<METRIC>(d, na, vmax, ctl->rho);
    float t, tv;
    if (METRIC == 0) { t = A / na - 2.02f * eps; tv = t * na; }
    else if (METRIC == 1) { t = (A - na_sq) - 2.02f * eps; tv = t + na_sq; }
    else { t = A - 2.02f * eps; tv = t; }
    if (!(t == t) || !(tv == tv) || !(tv > tv_cur)) return;   // nothing gained (or out of range): keep the list
    const uint32_t tkey = score_key(tv);
    uint32_t kept = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const uint32_t i = (uint32_t)tid + (uint32_t)e * PRUNE_THREADS;
        if (i < C && (uint32_t)(key[e] >> 32) >= tkey) {
            cq[above[e]] = row[e];   // (every thread read its rows before the first barrier)
            ++kept;
        }
    }
    if (kept) atomicAdd(&shared3[1], kept);
    __syncthreads();
    if (tid == 0) {
        ctl->cand_count = shared3[1];
        ctl->t_key = score_key(t);
        *thr_y = tv;
    }
}

// Lists longer than 512 entries (small batches, whose few workgroups raise the thresholds late, and
// clustered corpora): counting every entry against every other would cost more than the sort it
// replaced (1.25 M x 384, 256 queries, lists of ~800: 81 us against 28). Instead A is found by three
// radix levels over the keys in LDS, as the thresholds are, and only the survivors -- the rows at or
// above the lowered threshold, a few tens -- are ranked by counting. More than 512 survivors: the
// list is left as it is (it is still every row above the threshold the pass used).
constexpr uint32_t PRUNE_SURV = 512;
template <int METRIC>
__device__ __forceinline__ void prune_long_list(SearchCtl* __restrict__ ctl, uint32_t* __restrict__ cq,
                                                const uint32_t* __restrict__ vq, float* __restrict__ thr_y,
                                                uint32_t d, uint32_t C, uint32_t need, float na, float na_sq,
                                                float tv_cur, uint32_t* keys32, uint32_t* hist,
                                                unsigned long long* surv_key, uint32_t* surv_row,
                                                uint32_t* shared3, uint32_t* sh6) {
    const int tid = threadIdx.x;
    uint32_t n_inf = 0;
    for (uint32_t i = (uint32_t)tid; i < C; i += (uint32_t)PRUNE_THREADS) {
        const float v = __uint_as_float(vq[i]);
        const uint32_t k32 = (v != v || v == INFINITY) ? 0xFFFFFFFFu : score_key(v);
        n_inf += k32 == 0xFFFFFFFFu;
        keys32[i] = k32;
    }
    if (n_inf) atomicAdd(&shared3[0], n_inf);
    __syncthreads();
    const uint32_t F = shared3[0];
    if (F + need > C) return;   // fewer than `need` rows with a usable score: keep the whole list
    // the (F + need)-th largest key (the rows kept whatever the threshold carry the largest key)
    uint32_t want = F + need, prefix = 0, prefix_bits = 0;
    const int shifts[3] = {21, 10, 0};
    const uint32_t widths[3] = {11, 11, 10};
#pragma unroll
    for (int level = 0; level < 3; ++level) {
        for (int i = tid; i < 2048; i += PRUNE_THREADS) hist[i] = 0u;
        __syncthreads();
        const uint32_t mshift = 32u - prefix_bits;
        const uint32_t shift = (uint32_t)shifts[level], mask = (1u << widths[level]) - 1u;
        for (uint32_t i = (uint32_t)tid; i < C; i += (uint32_t)PRUNE_THREADS) {
            const uint32_t k32 = keys32[i];
            if (prefix_bits == 0 || (k32 >> mshift) == prefix) atomicAdd(&hist[(k32 >> shift) & mask], 1u);
        }
        __syncthreads();
        uint32_t bin, above;
        find_bin_256(hist, want, bin, above, sh6);
        want -= above;
        prefix = (prefix << widths[level]) | bin;
        prefix_bits += widths[level];
    }
    const float A = key_score(prefix);
    const float vmax = sqrtf(ctl->vmax_sq);
    const float eps = bf16_eps<METRIC>(d, na, vmax, ctl->rho);
    float t, tv;
    if (METRIC == 0) { t = A / na - 2.02f * eps; tv = t * na; }
    else if (METRIC == 1) { t = (A - na_sq) - 2.02f * eps; tv = t + na_sq; }
    else { t = A - 2.02f * eps; tv = t; }
    if (!(t == t) || !(tv == tv) || !(tv > tv_cur)) return;   // nothing gained (or out of range): keep the list
    const uint32_t tkey = score_key(tv);
    for (uint32_t i = (uint32_t)tid; i < C; i += (uint32_t)PRUNE_THREADS) {
        const uint32_t k32 = keys32[i];
        if (k32 >= tkey) {
            const uint32_t pos = atomicAdd(&shared3[1], 1u);
            if (pos < PRUNE_SURV) {
                surv_key[pos] = ((unsigned long long)k32 << 32) | (unsigned long long)(0xFFFFFFFFu - i);
                surv_row[pos] = cq[i];
            }
        }
    }
    __syncthreads();
    const uint32_t S = shared3[1];
    if (S > PRUNE_SURV) return;   // (nothing has been written yet)
    // every thread holds its survivors' rows before any is written back
    unsigned long long mine[2];
    uint32_t row[2], above2[2] = {0u, 0u};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const uint32_t i = (uint32_t)tid + (uint32_t)e * PRUNE_THREADS;
        mine[e] = i < S ? surv_key[i] : 0ull;
        row[e] = i < S ? surv_row[i] : 0u;
    }
    for (uint32_t j = 0; j < S; ++j) {
        const unsigned long long o = surv_key[j];
        above2[0] += o > mine[0] ? 1u : 0u;
        above2[1] += o > mine[1] ? 1u : 0u;
    }
#pragma unroll
    for (int e = 0; e < 2; ++e)
        if ((uint32_t)tid + (uint32_t)e * PRUNE_THREADS < S) cq[above2[e]] = row[e];
    if (tid == 0) {
        ctl->cand_count = S;
        ctl->t_key = score_key(t);
        *thr_y = tv;
    }
}

template <int METRIC>
__global__ void __launch_bounds__(PRUNE_THREADS)
qs_prune_kernel(SearchCtl* __restrict__ ctls, uint32_t* __restrict__ cand, const uint32_t* __restrict__ cvals,
                float* __restrict__ thr, uint32_t d, const uint32_t* __restrict__ dyn_lvl,
                const float* __restrict__ dyn_lad) {
    // short lists: 512 packed keys; long lists: 4096 keys, a histogram, the survivors' keys and rows
    __shared__ __attribute__((aligned(8))) uint32_t lds[CAND_CAP + 2048 + 2 * PRUNE_SURV + PRUNE_SURV];
    __shared__ uint32_t shared3[3];   // rows kept whatever the threshold, survivors, A's key
    __shared__ uint32_t sh6[6];
    static_assert(2 * 2 * PRUNE_THREADS <= CAND_CAP + 2048 + 3 * PRUNE_SURV, "short-list keys fit");
    const int tid = threadIdx.x;
    const uint32_t y = blockIdx.x;
    SearchCtl* ctl = ctls + y;
    const uint32_t C = ctl->cand_count;
    const uint32_t need = ctl->need;
    const float na = ctl->na, na_sq = ctl->na_sq;
    // score space (t: what the final check compares with) from v space (tv: what the pass compared with)
    auto to_score = [&](float tv) { return METRIC == 0 ? tv / na : (METRIC == 1 ? tv - na_sq : tv); };
    const uint32_t level = dyn_lvl[y];
    float tv_cur = thr[y];
    if (level > 0u) {
        tv_cur = dyn_level_value(level, dyn_lad[2 * y], dyn_lad[2 * y + 1]);
        if (tid == 0) {
            ctl->t_key = score_key(to_score(tv_cur));
            thr[y] = tv_cur;
        }
    }
#ifdef TRAG_LIST_DIAG
    if (tid == 0 && (y & 63u) == 0u) printf("query %u: list %u, level %u\n", y, C, level);
#endif
    if (C > CAND_CAP || C <= need) return;
    uint32_t* cq = cand + (size_t)y * CAND_CAP;
    const uint32_t* vq = cvals + (size_t)y * CAND_CAP;
    if (tid == 0) { shared3[0] = 0; shared3[1] = 0; }
    __syncthreads();
    unsigned long long* keys64 = reinterpret_cast<unsigned long long*>(lds);
    if (C <= (uint32_t)PRUNE_THREADS) prune_list<METRIC, 1>(ctl, cq, vq, thr + y, d, C, need, na, na_sq, tv_cur, keys64, shared3);
    else if (C <= 2u * PRUNE_THREADS) prune_list<METRIC, 2>(ctl, cq, vq, thr + y, d, C, need, na, na_sq, tv_cur, keys64, shared3);
    else
        prune_long_list<METRIC>(ctl, cq, vq, thr + y, d, C, need, na, na_sq, tv_cur, lds, lds + CAND_CAP,
                                reinterpret_cast<unsigned long long*>(lds + CAND_CAP + 2048),
                                lds + CAND_CAP + 2048 + 2 * PRUNE_SURV, shared3, sh6);
}

bool qs_supports(uint32_t dim, uint32_t metric) {
    return metric <= TRAG_DOT_PRODUCT && (dim == 128 || dim == 256 || dim == 384);
}

uint32_t qs_group_queries(uint32_t) { return 256u; }

#ifndef QS_MAX_STEP
#define QS_MAX_STEP 128
#endif
#ifndef QS_SAMPLE_GROUPS
#define QS_SAMPLE_GROUPS 16
#endif
double bf16_step_by_list(uint32_t need, uint32_t dim);   // trag_bf16.hip

// Most queries one pass takes: 8 query groups (the streams per XCD shrink with the groups).
uint32_t qs_max_batch(uint32_t dim) { return 8u * qs_group_queries(dim); }

// Rows per LDS tile for this dimension (the granularity of the sample pass).
uint32_t qs_tile_rows(uint32_t) { return 64u; }

template <int KS, int NW, int QB, int RT, int MODE, bool MIR, int DP, bool NTL>
static void qs_launch_tile(uint32_t n_queries, const ScanArgs& a, const uint16_t* d_mirror, const float* d_norm2,
                          const Bf16Scratch& s, uint32_t tile_step, uint32_t need, hipStream_t st) {
    static int cu_per_xcd = 0;
    if (cu_per_xcd == 0) {
        int dev = 0, n_cu = 256;
        if (hipGetDevice(&dev) == hipSuccess)
            (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
        cu_per_xcd = std::max(1, n_cu / 8);
    }
    const uint32_t QW = 32u * NW * QB;
    const uint32_t G = (n_queries + QW - 1) / QW;
    const uint32_t S = std::max<uint32_t>(1u, (uint32_t)cu_per_xcd / G);
    const uint32_t grid = 8u * S * G;
    hipLaunchKernelGGL((qs_gemm_kernel<KS, NW, QB, RT, MODE, MIR, DP, NTL>), dim3(grid), dim3(64 * NW), 0, st, a.corpus,
                       d_mirror, reinterpret_cast<const __bf16*>(s.qb), s.rowc, a.n_rows, n_queries, G, S,
                       tile_step, s.ctls, s.tmax, s.tmax_stride, s.cand, s.ckeys, need, s.dyn_lvl, s.dyn_lad,
                       s.dyn_cnt);
}

// The corpus is read by G workgroups of one XCD: keep it in L2 for them; a single reader (one
// query group) streams past the caches with non-temporal loads.
template <int KS, int NW, int QB, int RT, int MODE, bool MIR, int DP>
static void qs_launch_one(uint32_t n_queries, const ScanArgs& a, const uint16_t* d_mirror, const float* d_norm2,
                          const Bf16Scratch& s, uint32_t tile_step, uint32_t need, hipStream_t st) {
    if (n_queries <= 32u * NW * QB) qs_launch_tile<KS, NW, QB, RT, MODE, MIR, DP, true>(n_queries, a, d_mirror, d_norm2, s, tile_step, need, st);
    else qs_launch_tile<KS, NW, QB, RT, MODE, MIR, DP, false>(n_queries, a, d_mirror, d_norm2, s, tile_step, need, st);
}

#ifndef QS_DP_A
#define QS_DP_A 1
#endif
// d = 384: waves per workgroup (8: two per SIMD, one block of 32 queries each; 4: one per SIMD with
// the 512-register budget, two blocks each) and staging depth
#ifndef QS_NW_384
#define QS_NW_384 8
#endif
#ifndef QS_DP_384
#define QS_DP_384 QS_DP_A
#endif
template <int MODE, bool MIR>
static void qs_launch(uint32_t n_queries, const ScanArgs& a, const uint16_t* d_mirror, const float* d_norm2,
                      const Bf16Scratch& s, uint32_t tile_step, uint32_t need, hipStream_t st) {
    switch (a.dim) {
    case 128: qs_launch_one<8, 8, 1, 64, MODE, MIR, QS_DP_A>(n_queries, a, d_mirror, d_norm2, s, tile_step, need, st); break;
    case 256: qs_launch_one<16, 8, 1, 64, MODE, MIR, QS_DP_A>(n_queries, a, d_mirror, d_norm2, s, tile_step, need, st); break;
    case 384: qs_launch_one<24, QS_NW_384, 8 / QS_NW_384, 64, MODE, MIR, QS_DP_384>(n_queries, a, d_mirror, d_norm2, s, tile_step, need, st); break;
    default: break;   // (qs_supports: the caller never comes here with another dimension)
    }
}

// Tiles between two sampled tiles. As bf16_sample_step; the lists may be longer here (the prune
// stage shortens them before anything is re-scored), and the sample sparser: a pass over a 128th
// of the tiles still gives every stream a tile or two (1.25 M x 384 x 1000 queries, ms per step on one
// device: every 61st tile 1.260, every 122nd 1.242, every 244th 1.244).
uint32_t qs_sample_step(uint64_t n_rows, uint32_t need, uint32_t dim) {
    const uint64_t n_sub = (n_rows + 31) / 32;
    if (need == 0) need = 1;
    uint64_t step = n_sub / ((uint64_t)QS_SAMPLE_GROUPS * need);   // the sample must hold many more 32-row groups than rows wanted
    const double by_list = bf16_step_by_list(need, dim);
    if ((double)step > by_list) step = (uint64_t)by_list;
    return (uint32_t)(step < 1 ? 1 : (step > QS_MAX_STEP ? QS_MAX_STEP : step));
}

// declared in trag_bf16.hip
void launch_tau_shift(uint32_t metric, const Bf16Scratch& s, uint32_t n_maxima, uint32_t n_queries, uint32_t padded,
                      uint32_t dim, hipStream_t st);

void launch_qs_prune(uint32_t metric, const Bf16Scratch& s, uint32_t n_queries, uint32_t dim, hipStream_t st) {
    if (metric == TRAG_COSINE)
        hipLaunchKernelGGL(qs_prune_kernel<0>, dim3(n_queries), dim3(PRUNE_THREADS), 0, st, s.ctls, s.cand, s.ckeys, s.thr, dim, s.dyn_lvl, s.dyn_lad);
    else if (metric == TRAG_EUCLIDEAN)
        hipLaunchKernelGGL(qs_prune_kernel<1>, dim3(n_queries), dim3(PRUNE_THREADS), 0, st, s.ctls, s.cand, s.ckeys, s.thr, dim, s.dyn_lvl, s.dyn_lad);
    else
        hipLaunchKernelGGL(qs_prune_kernel<2>, dim3(n_queries), dim3(PRUNE_THREADS), 0, st, s.ctls, s.cand, s.ckeys, s.thr, dim, s.dyn_lvl, s.dyn_lad);
}

// The whole prefilter for up to qs_max_batch(dim) queries: sample pass, thresholds, full pass, prune.
// The caller has run launch_bf16_prep (norms, control blocks, bf16 queries, per-row constants).
hipError_t launch_qs_scan(uint32_t n_queries, const ScanArgs& a, const uint16_t* d_mirror,
                          const float* d_norm2, const Bf16Scratch& s, uint32_t need, hipStream_t st) {
    const uint32_t QW = qs_group_queries(a.dim);
    const uint32_t padded = (n_queries + QW - 1) / QW * QW;
    const uint32_t step = qs_sample_step(a.n_rows, need, a.dim);
    const uint32_t RT = qs_tile_rows(a.dim);
    if (d_mirror) qs_launch<0, true>(n_queries, a, d_mirror, d_norm2, s, step, need, st);
    else qs_launch<0, false>(n_queries, a, d_mirror, d_norm2, s, step, need, st);
    const uint32_t n_tiles = (a.n_rows + RT - 1) / RT;
    const uint32_t sampled = (n_tiles + step - 1) / step;
    launch_tau_shift(a.metric, s, sampled * (RT / 32), n_queries, padded, a.dim, st);
    if (d_mirror) qs_launch<1, true>(n_queries, a, d_mirror, d_norm2, s, 1, need, st);
    else qs_launch<1, false>(n_queries, a, d_mirror, d_norm2, s, 1, need, st);
    launch_qs_prune(a.metric, s, n_queries, a.dim, st);
    return hipGetLastError();
}

}  // namespace trag
