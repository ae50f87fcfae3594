// Batched queries on gfx950: Q x D^T on the fp32 matrix cores with a fused
// scoring epilogue, then per-query threshold selection from tile maxima.
//
// For B queries the similarity matrix is a genuine GEMM (B x d times d x N),
// compute-bound for B >~ 40 (2*B*N*d flops over N*d*4 bytes). The products are
// exact fp32 (v_mfma_f32_32x32x2_f32 is a chain of fmaf), but the summation
// order differs from the reference's left-to-right mul+add, so the scores
// written here are approximations: as in the streaming single-query mode the
// final stage re-scores the selected candidates with the reference arithmetic
// and certifies the result against a worst-case bound.
#include "trag_device.h"
#include "trag_internal.h"

namespace trag {

typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int BT_KMIN = 32;      // dimensions must be a multiple of this

// prep for one query, by one workgroup of 256 threads: ||q|| in the reference's order, the query's
// control block reset, and (qb_row != nullptr) the query converted to bf16 for the prefilter.
__device__ __forceinline__ void prep_query(const float* __restrict__ q, uint32_t d, SearchCtl* __restrict__ ctl,
                                           uint32_t need, uint32_t k, const float* __restrict__ vmax_sq,
                                           __bf16* __restrict__ qb_row) {
    __shared__ __attribute__((aligned(16))) float qs[4096];
    __shared__ float err_sh[4];
    const int tid = threadIdx.x;
    // (asked for here, not where thread 0 uses them after the sum: one round trip less at the end)
    const float vmax0 = vmax_sq ? vmax_sq[0] : 0.0f, vmax1 = vmax_sq ? vmax_sq[1] : 1.0f;
    float acc = 0.0f;
    float e2 = 0.0f;   // sum of (bf16(q) - q)^2, for the bf16 prefilter's bound (any order)
    for (uint32_t base = 0; base < d; base += 4096) {
        const uint32_t len = min(4096u, d - base);
        __syncthreads();
        for (uint32_t i = tid; i < len; i += 256) {
            const float x = q[base + i];
            qs[i] = x;
            e2 = bf16_round_err_sq(x, e2);
            if (qb_row) qb_row[base + i] = (__bf16)x;
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t i = 0;
            // (unrolled so that the LDS reads of eight steps are in flight together: one thread alone
            // otherwise waits out an LDS latency per four components)
#pragma unroll 8
            for (; i + 4 <= len; i += 4) {
                const v4f x = *reinterpret_cast<const v4f*>(&qs[i]);
                float p0 = x.x * x.x, p1 = x.y * x.y, p2 = x.z * x.z, p3 = x.w * x.w;
                acc = acc + p0;
                acc = acc + p1;
                acc = acc + p2;
                acc = acc + p3;
            }
            for (; i < len; ++i) {
                float p = qs[i] * qs[i];
                acc = acc + p;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) e2 += __shfl_xor(e2, o);
    if ((tid & 63) == 0) err_sh[tid >> 6] = e2;
    __syncthreads();
    if (tid == 0) {
        ctl->na_sq = acc;
        ctl->na = sqrtf(acc);
        ctl->vmax_sq = vmax0;
        ctl->rho = bf16_operand_rho((err_sh[0] + err_sh[1]) + (err_sh[2] + err_sh[3]), acc, vmax1);
        ctl->need = need;
        ctl->k = k;
        ctl->resolved = 1;  // the threshold comes from the tile maxima (tau_select_kernel / tau_shift_kernel), not from histograms
        ctl->b1 = ctl->above1 = ctl->b2 = ctl->above2 = 0;
        ctl->t_key = 0;
        ctl->cut_ties = 0;
        ctl->tie_quota = 0;
        ctl->above3 = 0;
        ctl->expect = 0;
        ctl->cand_count = 0;
        ctl->status = ST_OK;
    }
}

// prep for a block of queries: one workgroup per query.
__global__ void __launch_bounds__(256)
prep_batch_kernel(const float* __restrict__ Q, uint32_t d, SearchCtl* __restrict__ ctls,
                  uint32_t need, uint32_t k, const float* __restrict__ vmax_sq) {
    prep_query(Q + (size_t)blockIdx.x * d, d, ctls + blockIdx.x, need, k, vmax_sq, nullptr);
}

// Everything the bf16 prefilter needs before its first pass, in one launch (three small launches in
// a row cost their latencies one after the other; here the per-row constants fill the chip while
// the query workgroups wait for the one thread that sums in the reference's order). Workgroups
// [0, n_queries): prep_query, which also writes the query's bf16 image; [n_queries, padded): zero
// rows behind the batch (the passes read whole query groups); the rest, if rowc != nullptr: (scale,
// shift) of every row up to n_pad, the next multiple of 64 rows (the query-stationary pass reads that far).
__global__ void __launch_bounds__(256)
bf16_prep_kernel(const float* __restrict__ Q, uint32_t d, SearchCtl* __restrict__ ctls, uint32_t need, uint32_t k,
                 const float* __restrict__ vmax_sq, uint32_t n_queries, uint32_t padded, __bf16* __restrict__ qb,
                 const float* __restrict__ norm2, const uint32_t* __restrict__ dead, uint32_t n_rows, uint32_t n_pad,
                 uint32_t metric, float2* __restrict__ rowc) {
    const uint32_t b = blockIdx.x;
    if (b < n_queries) {
        prep_query(Q + (size_t)b * d, d, ctls + b, need, k, vmax_sq, qb + (size_t)b * d);
    } else if (b < padded) {
        __bf16* row = qb + (size_t)b * d;
        for (uint32_t i = threadIdx.x; i < d; i += 256) row[i] = (__bf16)0.0f;
    } else {
        const uint32_t nb = gridDim.x - padded;
        for (uint32_t row = (b - padded) * 256u + threadIdx.x; row < n_pad; row += nb * 256u)
            rowc[row] = row_scale_shift(norm2, dead, n_rows, metric, row);
    }
}

// ---------------------------------------------------------------------------
// GEMM scan. Four waves per workgroup, arranged WGM (queries) x 4/WGM (rows);
// each wave owns TM x 2 MFMA tiles of 32 x 32, i.e. 32*TM queries x 64 rows.
//   <TM 2, WGM 2>: workgroup tile 128 queries x 128 rows (large batches, MFMA-bound)
//   <TM 2, WGM 1>:                 64 queries x 256 rows
//   <TM 1, WGM 1>:                 32 queries x 256 rows (few queries: the pass
//                  is then bound by the corpus read, not by padded tile flops)
// MFMA roles: A = queries (tile rows), B = corpus rows (tile columns), so an
// accumulator register holds one query for 32 consecutive corpus rows across
// the lanes and the key stores are 128-byte runs.
// Operands of v_mfma_f32_32x32x2_f32 are one float per lane:
//   a = A[i = lane & 31][k = lane >> 5],  b = B[k = lane >> 5][j = lane & 31].
// The 32 k-values of a staged step are assigned to the 16 MFMAs as
// k = 16 * (lane >> 5) + s (s = MFMA index), the same for A and B, so each lane
// reads its 16 operands as four contiguous ds_read_b128.
// METRIC: 0 cosine, 2 dot product, 1 Euclidean (keys order the negated squared
// distance ||q||^2 + ||v||^2 - 2 q.v from the cached norms; the final stage
// re-scores with the reference's sqrt(sum (x - y)^2) and certifies in squared space).
// ---------------------------------------------------------------------------
template <int METRIC, int BT_K /* k-step staged through LDS */, int TM, int WGM>
__global__ void __launch_bounds__(256)
gemm_scan_kernel(const float* __restrict__ corpus, const float* __restrict__ Qpad,
                 const float* __restrict__ norm2, const uint32_t* __restrict__ dead,
                 uint32_t n_rows, uint32_t d, uint32_t n_queries, uint32_t q_tiles,
                 const SearchCtl* __restrict__ ctls, uint32_t* __restrict__ keys,
                 size_t keys_stride, uint32_t* __restrict__ tmax, size_t tmax_stride) {
    constexpr int BT_LD = BT_K + 4;  // padded LDS row stride (floats): conflict-free b128 reads
    constexpr int WGN = 4 / WGM;
    constexpr int BT_M = 32 * TM * WGM;  // queries per workgroup tile
    constexpr int BT_N = 64 * WGN;       // corpus rows per workgroup tile
    constexpr int NLA = BT_M * BT_K / 4 / 256;  // 16-byte loads per thread for the query tile
    constexpr int NLB = BT_N * BT_K / 4 / 256;  // ... for the corpus tile
    constexpr int CPR = BT_K / 4;    // float4 columns per tile row
    constexpr int RPP = 256 / CPR;   // rows covered by one pass of the 256 threads
    constexpr int KH = BT_K / 2;     // k-values per lane half
    __shared__ __attribute__((aligned(16))) float As_[1][BT_M * BT_LD];
    __shared__ __attribute__((aligned(16))) float Bs_[1][BT_N * BT_LD];
    __shared__ float na_s[BT_M];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid % WGN;

    // Workgroups that share a corpus tile (different query tiles) get ids that
    // are equal mod 8 and adjacent in dispatch order, so that the second read
    // of the tile is served by the same XCD's L2.
    const uint32_t b = blockIdx.x;
    const uint32_t xcd = b & 7u, i8 = b >> 3;
    const uint32_t qt = i8 % q_tiles;
    const uint32_t nt = (i8 / q_tiles) * 8u + xcd;
    const uint32_t n_row_tiles = (n_rows + BT_N - 1) / BT_N;
    if (nt >= n_row_tiles) return;
    const uint32_t q0 = qt * BT_M, n0 = nt * BT_N;

    if (tid < BT_M)
        na_s[tid] = (q0 + tid < n_queries) ? (METRIC == 1 ? ctls[q0 + tid].na_sq : ctls[q0 + tid].na) : 0.0f;

    // staging map: thread -> (row = tid/CPR + RPP*i, float4 column = tid%CPR)
    const int srow = tid / CPR, scol = tid % CPR;
    const float* aptr[NLA];
    const float* bptr[NLB];
#pragma unroll
    for (int i = 0; i < NLA; ++i) aptr[i] = Qpad + (size_t)(q0 + srow + RPP * i) * d + scol * 4;
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
        const uint32_t r = min(n0 + (uint32_t)(srow + RPP * i), n_rows - 1u);
        bptr[i] = corpus + (size_t)r * d + scol * 4;
    }
    v4f ra[NLA], rb[NLB];
#pragma unroll
    for (int i = 0; i < NLA; ++i) ra[i] = *reinterpret_cast<const v4f*>(aptr[i]);
#pragma unroll
    for (int i = 0; i < NLB; ++i) rb[i] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(bptr[i]));

    v16f acc[TM][2];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

    const int frow = lane & 31, fk = (lane >> 5) * KH;
    const uint32_t nk = d / BT_K;
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLA; ++i)
            *reinterpret_cast<v4f*>(&As_[buf][(srow + RPP * i) * BT_LD + scol * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < NLB; ++i)
            *reinterpret_cast<v4f*>(&Bs_[buf][(srow + RPP * i) * BT_LD + scol * 4]) = rb[i];
    };
    auto prefetch = [&](uint32_t kt) {
#pragma unroll
        for (int i = 0; i < NLA; ++i) ra[i] = *reinterpret_cast<const v4f*>(aptr[i] + (size_t)kt * BT_K);
#pragma unroll
        for (int i = 0; i < NLB; ++i)
            rb[i] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(bptr[i] + (size_t)kt * BT_K));
    };
    auto compute = [&](int buf) {
        const float* As = As_[buf];
        const float* Bs = Bs_[buf];
#pragma unroll
        for (int kk = 0; kk < KH / 16; ++kk) {  // 16 MFMA k-slots per pass
            v4f fa[TM][4], fb[2][4];
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    fa[m][j] = *reinterpret_cast<const v4f*>(
                        &As[(wm * 32 * TM + m * 32 + frow) * BT_LD + fk + kk * 16 + j * 4]);
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    fb[n][j] = *reinterpret_cast<const v4f*>(
                        &Bs[(wn * 64 + n * 32 + frow) * BT_LD + fk + kk * 16 + j * 4]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int m = 0; m < TM; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[m][j][e], fb[n][j][e],
                                                                             acc[m][n], 0, 0, 0);
        }
    };
    // Measured alternatives on MI355X (1M x 768, 256 queries): this loop 107-111 TFLOP/s;
    // two LDS images with one barrier per step 103; 64-wide k-steps 102; s_setprio
    // around the MFMA block 106. The chip holds ~2.03 GHz under this load.
    for (uint32_t kt = 0; kt < nk; ++kt) {
        __syncthreads();
        stage(0);
        __syncthreads();
        if (kt + 1 < nk) prefetch(kt + 1);
        compute(0);
    }

    // ---- epilogue: scores -> ordered keys, per-(query, 64-row tile) maxima ----
    const int half = lane >> 5;
    uint32_t rows_[2];
    float inv_nb[2], nb2_[2];
    bool live[2], ok_nb[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const uint32_t row = n0 + wn * 64 + n * 32 + (lane & 31);
        rows_[n] = row;
        live[n] = row < n_rows && !row_is_dead(dead, row);
        float nb2 = 1.0f;
        ok_nb[n] = true;
        if (METRIC != 2) {
            nb2 = row < n_rows ? norm2[row] : 0.0f;
            // outside this range the error bound of the certification does not hold
            ok_nb[n] = nb2 == 0.0f || (nb2 >= 0x1p-100f && nb2 < INFINITY);
        }
        inv_nb[n] = nb2 > 0.0f ? 1.0f / sqrtf(nb2) : 0.0f;
        nb2_[n] = nb2;
    }
    const uint32_t tile64 = (n0 + wn * 64) >> 6;
#pragma unroll
    for (int m = 0; m < TM; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qi = wm * 32 * TM + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const uint32_t q = q0 + qi;
            const float na = na_s[qi];
            uint32_t kmax = 0u;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const float dot = acc[m][n][r];
                float s;
                if (METRIC == 0) {
                    s = (na > 0.0f) ? dot * (inv_nb[n] / na) : 0.0f;
                    if (!ok_nb[n]) s = INFINITY;
                } else if (METRIC == 1) {
                    const float two_dot = 2.0f * dot;
                    const float sum2 = na + nb2_[n];   // na holds ||q||^2 for this metric
                    s = -(sum2 - two_dot);
                    if (!ok_nb[n]) s = INFINITY;
                } else {
                    s = dot;
                }
                if (!(fabsf(dot) < INFINITY)) s = INFINITY;  // overflow / NaN: always re-scored
                const uint32_t key = live[n] ? score_key(s) : 0u;
                if (q < n_queries && rows_[n] < n_rows) keys[q * keys_stride + rows_[n]] = key;
                kmax = key > kmax ? key : kmax;
            }
            // max over the 32 lanes of this half (they hold the same query)
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) {
                const uint32_t w = (uint32_t)__shfl_xor((int)kmax, o);
                kmax = w > kmax ? w : kmax;
            }
            if ((lane & 31) == 0 && q < n_queries && (uint32_t)(n0 + wn * 64) < n_rows)
                tmax[q * tmax_stride + tile64] = kmax;
        }
    }
}

// ---------------------------------------------------------------------------
// Per-query threshold: the need-th largest tile maximum. At least `need` tiles
// hold a key at or above it, hence at least `need` rows do, so the rows with
// key >= tau contain the top `need`; because the best rows of a large corpus
// almost always sit in different tiles, tau is close to the need-th best key.
// One workgroup per query; three radix levels (11 + 11 + 10 bits) over the
// query's tile maxima with workgroup-local histograms.
// ---------------------------------------------------------------------------
__device__ void find_bin_lds(const uint32_t* hist, uint32_t need, uint32_t& bin, uint32_t& above,
                             uint32_t* sh);  // defined below

__global__ void __launch_bounds__(1024)
tau_select_kernel(const uint32_t* __restrict__ tmax, size_t tmax_stride, uint32_t n_tiles,
                  SearchCtl* __restrict__ ctls) {
    __shared__ uint32_t lhist[HIST_BINS];
    __shared__ uint32_t sh[18];
    const int tid = threadIdx.x;
    const uint32_t* tm = tmax + (size_t)blockIdx.x * tmax_stride;
    SearchCtl* ctl = ctls + blockIdx.x;
    uint32_t need = ctl->need;
    if (need > n_tiles) {  // fewer tiles than wanted rows: take every live row
        if (tid == 0) ctl->t_key = 1u;
        return;
    }
    uint32_t prefix = 0, prefix_bits = 0;
    const int shifts[3] = {21, 10, 0};
    const uint32_t widths[3] = {11, 11, 10};
    for (int level = 0; level < 3; ++level) {
        for (int i = tid; i < (int)HIST_BINS; i += 1024) lhist[i] = 0u;
        __syncthreads();
        const uint32_t mshift = 32u - prefix_bits;
        const uint32_t shift = (uint32_t)shifts[level], mask = (1u << widths[level]) - 1u;
        // four independent loads in flight per thread: the pass is latency-bound
        for (uint32_t i0 = 0; i0 < n_tiles; i0 += 4096u) {
            uint32_t v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t i = i0 + (uint32_t)j * 1024u + (uint32_t)tid;
                v[j] = i < n_tiles ? tm[i] : 0u;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t i = i0 + (uint32_t)j * 1024u + (uint32_t)tid;
                const bool match = prefix_bits == 0 || (v[j] >> mshift) == prefix;
                if (i < n_tiles && match) atomicAdd(&lhist[(v[j] >> shift) & mask], 1u);
            }
        }
        __syncthreads();
        uint32_t bin, above;
        find_bin_lds(lhist, need, bin, above, sh);
        need -= above;
        prefix = (prefix << widths[level]) | bin;
        prefix_bits += widths[level];
    }
    if (tid == 0) ctl->t_key = prefix > 1u ? prefix : 1u;
}

// find_bin over a histogram in LDS, called by all 1024 threads of the workgroup:
// the highest bin whose suffix count reaches `need`, and the count strictly above
// it. sh: 18 words (16 wave totals, then bin and count).
__device__ void find_bin_lds(const uint32_t* hist, uint32_t need, uint32_t& bin, uint32_t& above,
                             uint32_t* sh) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int top = (int)HIST_BINS - 1 - 2 * t;  // thread t owns bins top and top - 1
    const uint32_t h0 = hist[top], h1 = hist[top - 1];
    const uint32_t sum = h0 + h1;
    uint32_t incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) sh[w] = incl;
    if (t == 0) { sh[16] = 0; sh[17] = 0; }
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (int i = 0; i < 16; ++i) {
        if (i < w) base += sh[i];
        total += sh[i];
    }
    incl += base;
    const uint32_t excl = incl - sum;
    if (need > total) need = total;
    if (need > 0 && excl < need && need <= incl) {
        if (excl + h0 >= need) { sh[16] = (uint32_t)top; sh[17] = excl; }
        else { sh[16] = (uint32_t)(top - 1); sh[17] = excl + h0; }
    }
    __syncthreads();
    bin = sh[16];
    above = sh[17];
    __syncthreads();
}

// Zero-padded copy of the query block (rows beyond n_queries are zero) so that
// the GEMM tiles never read outside the buffer.
__global__ void __launch_bounds__(256)
pad_queries_kernel(const float* __restrict__ Q, uint32_t n_queries, uint32_t padded, uint32_t d,
                   float* __restrict__ out) {
    const size_t total = (size_t)padded * d;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t q = e / d;
        out[e] = q < n_queries ? Q[e] : 0.0f;
    }
}

// with_rowc: also the per-row (scale, shift) table of the query-stationary pass.
hipError_t launch_bf16_prep(const float* d_Q, uint32_t n_queries, uint32_t padded, const ScanArgs& a,
                            const float* d_norm2, const Bf16Scratch& s, uint32_t need, uint32_t k,
                            const float* d_vmax_sq, bool with_rowc, hipStream_t st) {
    const uint32_t n_pad = (a.n_rows + 63u) / 64u * 64u;
    const uint32_t row_groups = with_rowc ? std::min<uint32_t>((n_pad + 255u) / 256u, 2048u) : 0u;
    hipLaunchKernelGGL(bf16_prep_kernel, dim3(padded + row_groups), dim3(256), 0, st, d_Q, a.dim, s.ctls, need, k,
                       d_vmax_sq, n_queries, padded, reinterpret_cast<__bf16*>(s.qb), d_norm2, a.dead, a.n_rows, n_pad,
                       a.metric, reinterpret_cast<float2*>(with_rowc ? s.rowc : nullptr));
    return hipGetLastError();
}

bool batch_supports(uint32_t dim, uint32_t metric) {
    return (dim % BT_KMIN) == 0 && metric <= TRAG_DOT_PRODUCT;
}

hipError_t launch_batch_scan(const float* d_Q, uint32_t n_queries, const ScanArgs& a,
                             const float* d_norm2, const BatchScratch& s, uint32_t need, uint32_t k,
                             const float* d_vmax_sq, hipStream_t st) {
    // query-tile height by batch size: a short batch must not pay for a 128-query tile
    const int variant = n_queries <= 32 ? 0 : (n_queries <= 64 ? 1 : 2);
    const uint32_t BT_M = variant == 0 ? 32 : (variant == 1 ? 64 : 128);
    const uint32_t BT_N = variant == 2 ? 128 : 256;
    const uint32_t q_tiles = (n_queries + BT_M - 1) / BT_M;
    const uint32_t padded = q_tiles * BT_M;
    hipLaunchKernelGGL(pad_queries_kernel, dim3(std::min<uint32_t>(1024, (padded * a.dim + 255) / 256)),
                       dim3(256), 0, st, d_Q, n_queries, padded, a.dim, s.qpad);
    hipLaunchKernelGGL(prep_batch_kernel, dim3(n_queries), dim3(256), 0, st, d_Q, a.dim, s.ctls, need,
                       k, d_vmax_sq);
    const uint32_t n_row_tiles = (a.n_rows + BT_N - 1) / BT_N;
    const uint32_t groups = (n_row_tiles + 7) / 8;  // corpus tiles in groups of 8 (one per XCD slot)
    const uint32_t grid = groups * q_tiles * 8;
#define TRAG_LAUNCH_GEMM(METRIC, TM, WGM)                                                              \
    hipLaunchKernelGGL((gemm_scan_kernel<METRIC, 32, TM, WGM>), dim3(grid), dim3(256), 0, st, a.corpus, \
                       s.qpad, d_norm2, a.dead, a.n_rows, a.dim, n_queries, q_tiles, s.ctls, s.keys,    \
                       s.keys_stride, s.tmax, s.tmax_stride)
    if (a.metric == TRAG_COSINE) {
        if (variant == 0) TRAG_LAUNCH_GEMM(0, 1, 1);
        else if (variant == 1) TRAG_LAUNCH_GEMM(0, 2, 1);
        else TRAG_LAUNCH_GEMM(0, 2, 2);
    } else if (a.metric == TRAG_EUCLIDEAN) {
        if (variant == 0) TRAG_LAUNCH_GEMM(1, 1, 1);
        else if (variant == 1) TRAG_LAUNCH_GEMM(1, 2, 1);
        else TRAG_LAUNCH_GEMM(1, 2, 2);
    } else {
        if (variant == 0) TRAG_LAUNCH_GEMM(2, 1, 1);
        else if (variant == 1) TRAG_LAUNCH_GEMM(2, 2, 1);
        else TRAG_LAUNCH_GEMM(2, 2, 2);
    }
#undef TRAG_LAUNCH_GEMM
    const uint32_t n_tiles64 = (a.n_rows + 63) / 64;
    hipLaunchKernelGGL(tau_select_kernel, dim3(n_queries), dim3(1024), 0, st, s.tmax, s.tmax_stride,
                       n_tiles64, s.ctls);
    return hipGetLastError();
}

}  // namespace trag
