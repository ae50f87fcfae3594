// EXPERIMENT, NOT PART OF THE LIBRARY: a timing probe for the k-loop of a 256 x 256 bf16 tile computed by FOUR waves
// (one per SIMD, 128 queries x 128 rows each = 16 accumulator tiles of v_mfma_f32_32x32x16_bf16, 8 fragment reads
// per 16 MFMAs), both operands brought into LDS by direct-to-LDS transfers, two stages, ONE barrier per k-step placed
// behind the step's last fragment reads. No threshold test, no hits: the accumulators are summed and stored.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/experiments/w4_probe.hip -o gpurun_out/w4_probe
//   w4_probe <rows> <dim> [reps]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#ifndef W4_EPI
#define W4_EPI 0
#endif

constexpr int BM = 256, BN = 256, BF_K = 64;
constexpr int STAGE_DW = (BM + BN) * 32;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(256)
w4_kernel(const uint16_t* __restrict__ mirror, const uint16_t* __restrict__ Qb, uint32_t n_rows, uint32_t d,
          float* __restrict__ out, const float* __restrict__ thr, uint32_t* __restrict__ hits,
          unsigned long long* __restrict__ clk) {
    __shared__ __attribute__((aligned(1024))) uint32_t lds0[STAGE_DW];
    __shared__ __attribute__((aligned(1024))) uint32_t lds1[STAGE_DW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const uint32_t n0 = blockIdx.x * BN;
    const uint32_t nk = d / BF_K;
    const int frow = lane & 31, half = lane >> 5;
    const int lrow = lane >> 3, slot = lane & 7;
    constexpr int NA = BM / 8 / 4, NB = BN / 8 / 4;   // 8-row blocks per wave and step
    uint32_t voff_a[2], voff_b[NB];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = i * 8 + lrow;
        voff_a[i] = (uint32_t)lrow * d * 2u + (uint32_t)((slot ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int row = (wid * NB + i) * 8 + lrow;
        uint32_t r = n0 + (uint32_t)row;
        if (r > n_rows - 1u) r = n_rows - 1u;
        voff_b[i] = (r - n0) * d * 2u + (uint32_t)((slot ^ ((row >> 1) & 7)) << 4);
    }
    const char* const a_base = reinterpret_cast<const char*>(Qb) + (size_t)(wid * NA * 8) * d * 2u;
    const char* const b_base = reinterpret_cast<const char*>(mirror) + (size_t)n0 * d * 2u;
    typedef __attribute__((address_space(3))) uint32_t* lds_u32;
    auto transfer = [&](uint32_t* dst, uint32_t kt) __attribute__((always_inline)) {
        const uint32_t m_a = (uint32_t)(uintptr_t)(lds_u32)(dst) + (uint32_t)(wid * NA) * 1024u;
        const uint32_t m_b = (uint32_t)(uintptr_t)(lds_u32)(dst + BM * 32) + (uint32_t)(wid * NB) * 1024u;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const char* sb = a_base + (size_t)(i * 8) * d * 2u + (size_t)kt * (BF_K * 2);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m_a + (uint32_t)i * 1024u), "v"(voff_a[i & 1]), "s"(sb) : "memory");
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const char* sb = b_base + (size_t)kt * (BF_K * 2);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt" ::"s"(m_b + (uint32_t)i * 1024u), "v"(voff_b[i]), "s"(sb) : "memory");
        }
    };
    v16f acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
        {
            // (zeroed by the matrix pipe itself, 0 x 0 + 0: a vector constant would pass through 16 vector registers)
            const bf16x8 z = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %1, 0" : "=a"(acc[m][n]) : "v"(z));
        }
    const int sw = (frow >> 1) & 7;
    bf16x8 fa[2][4], fb[2][4];
    auto rd_a = [&](const uint32_t* st, int ks, int m) __attribute__((always_inline)) {
        const int chunk = ((ks * 2 + half) ^ sw) << 2;
        return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&st[(wm * 128 + m * 32 + frow) * 32 + chunk]));
    };
    auto rd_b = [&](const uint32_t* st, int ks, int n) __attribute__((always_inline)) {
        const int chunk = ((ks * 2 + half) ^ sw) << 2;
        return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&st[BM * 32 + (wn * 128 + n * 32 + frow) * 32 + chunk]));
    };
    // one k-step: MFMA steps 0..3 out of `cur`; the fragments of MFMA step ks + 1 are read two by two behind the four
    // MFMAs of each query block of step ks; behind the reads of step 3 the barrier (everyone is done with `cur`, and
    // everyone's transfers of the next k-step have landed in `nxt`), then the transfers of the k-step after next go
    // into `cur` and the first fragments of the next k-step are read from `nxt` behind the MFMAs of step 3.
    auto kstep = [&](const uint32_t* __restrict__ cur, uint32_t* __restrict__ nxt, bool more1, bool more2, uint32_t kt)
                     __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cb = ks & 1, nb = (ks + 1) & 1;
            if (ks == 3) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifndef W4_NOXFER
                if (more2) transfer(const_cast<uint32_t*>(cur), kt + 2u);
#endif
            }
            const uint32_t* src = ks == 3 ? nxt : cur;
            const int nks = ks == 3 ? 0 : ks + 1;
            const bool rd = ks < 3 || more1;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
#pragma unroll
                for (int n = 0; n < 4; ++n)
#ifndef W4_NOMFMA
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[m][n]) : "v"(fa[cb][m]), "v"(fb[cb][n]));
#else
                    asm volatile("; %0 %1 %2" : "+a"(acc[m][n]) : "v"(fa[cb][m]), "v"(fb[cb][n]));
#endif
                __builtin_amdgcn_sched_barrier(0);
#ifdef W4_NORD
                if (false) {
#else
                if (rd) {
#endif
                    // order of the eight reads: b0 a0 | b1 b2 | b3 a1 | a2 a3 -- each at least seven MFMAs ahead of its use
                    if (m == 0) { fb[nb][0] = rd_b(src, nks, 0); fa[nb][0] = rd_a(src, nks, 0); }
                    if (m == 1) { fb[nb][1] = rd_b(src, nks, 1); fb[nb][2] = rd_b(src, nks, 2); }
                    if (m == 2) { fb[nb][3] = rd_b(src, nks, 3); fa[nb][1] = rd_a(src, nks, 1); }
                    if (m == 3) { fa[nb][2] = rd_a(src, nks, 2); fa[nb][3] = rd_a(src, nks, 3); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // (in-kernel clock: shader cycles per 100 MHz tick of the constant clock, around the k-loop)
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    transfer(lds0, 0u);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (nk > 1u) transfer(lds1, 1u);
#pragma unroll
    for (int i = 0; i < 4; ++i) { fb[0][i] = rd_b(lds0, 0, i); fa[0][i] = rd_a(lds0, 0, i); }
    for (uint32_t kt = 0; kt < nk; kt += 2) {
        kstep(lds0, lds1, kt + 1u < nk, kt + 2u < nk, kt);
        if (kt + 1u < nk) kstep(lds1, lds0, kt + 2u < nk, kt + 3u < nk, kt + 1u);
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // (the last MFMAs' results, read by vector instructions below)
    if (tid == 0 && (blockIdx.x & 63u) == 7u) {
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        clk[2 * (blockIdx.x >> 6)] = c1 - c0;
        clk[2 * (blockIdx.x >> 6) + 1] = r1 - r0;
    }
#if W4_EPI == 0
    float s = 0.0f;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[m][n][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
#else
    // the threshold test as the library's full pass does it: one compare per accumulator against the query's threshold
    uint32_t nh = 0;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float th = thr[wm * 128 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
            bool h = false;
#pragma unroll
            for (int n = 0; n < 4; ++n) h |= !(acc[m][n][r] < th);
            if (__builtin_amdgcn_ballot_w64(h)) nh += h ? 1u : 0u;
        }
    if (nh) atomicAdd(hits, nh);
    if (tid == 0 && blockIdx.x == 0) out[0] = acc[0][0][0];
#endif
}


// The same tile by v_mfma_f32_16x16x32_bf16: 8 x 8 output tiles of 16 x 16 per wave (4 accumulator registers each), per 32
// columns 8 + 8 fragment reads for 64 MFMAs of 16 cycles.
typedef float v4f_ __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256)
w4_kernel16(const uint16_t* __restrict__ mirror, const uint16_t* __restrict__ Qb, uint32_t n_rows, uint32_t d,
            float* __restrict__ out, const float* __restrict__ thr, uint32_t* __restrict__ hits,
            unsigned long long* __restrict__ clk) {
    __shared__ __attribute__((aligned(1024))) uint32_t lds0[STAGE_DW];
    __shared__ __attribute__((aligned(1024))) uint32_t lds1[STAGE_DW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const uint32_t n0 = blockIdx.x * BN;
    const uint32_t nk = d / BF_K;
    const int frow = lane & 15, quad = lane >> 4;
    const int lrow = lane >> 3, slot = lane & 7;
    constexpr int NA = BM / 8 / 4, NB = BN / 8 / 4;
    uint32_t voff_a[2], voff_b[NB];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = i * 8 + lrow;
        voff_a[i] = (uint32_t)lrow * d * 2u + (uint32_t)((slot ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int row = (wid * NB + i) * 8 + lrow;
        uint32_t r = n0 + (uint32_t)row;
        if (r > n_rows - 1u) r = n_rows - 1u;
        voff_b[i] = (r - n0) * d * 2u + (uint32_t)((slot ^ ((row >> 1) & 7)) << 4);
    }
    const char* const a_base = reinterpret_cast<const char*>(Qb) + (size_t)(wid * NA * 8) * d * 2u;
    const char* const b_base = reinterpret_cast<const char*>(mirror) + (size_t)n0 * d * 2u;
    typedef __attribute__((address_space(3))) uint32_t* lds_u32;
    auto transfer = [&](uint32_t* dst, uint32_t kt) __attribute__((always_inline)) {
        const uint32_t m_a = (uint32_t)(uintptr_t)(lds_u32)(dst) + (uint32_t)(wid * NA) * 1024u;
        const uint32_t m_b = (uint32_t)(uintptr_t)(lds_u32)(dst + BM * 32) + (uint32_t)(wid * NB) * 1024u;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const char* sb = a_base + (size_t)(i * 8) * d * 2u + (size_t)kt * (BF_K * 2);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m_a + (uint32_t)i * 1024u), "v"(voff_a[i & 1]), "s"(sb) : "memory");
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const char* sb = b_base + (size_t)kt * (BF_K * 2);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt" ::"s"(m_b + (uint32_t)i * 1024u), "v"(voff_b[i]), "s"(sb) : "memory");
        }
    };
    v4f_ acc[8][8];
    {
        const bf16x8 z = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 8; ++n) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, 0" : "=a"(acc[m][n]) : "v"(z));
    }
    // rows (m * 16 + frow) differ from frow by multiples of 16, so the swizzle term is that of frow
    const int sw = (frow >> 1) & 7;
    bf16x8 fa[2][8], fb[2][8];
    auto rd_a = [&](const uint32_t* st, int ks, int m) __attribute__((always_inline)) {
        const int chunk = ((ks * 4 + quad) ^ sw) << 2;
        return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&st[(wm * 128 + m * 16 + frow) * 32 + chunk]));
    };
    auto rd_b = [&](const uint32_t* st, int ks, int n) __attribute__((always_inline)) {
        const int chunk = ((ks * 4 + quad) ^ sw) << 2;
        return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&st[BM * 32 + (wn * 128 + n * 16 + frow) * 32 + chunk]));
    };
    auto kstep = [&](const uint32_t* __restrict__ cur, uint32_t* __restrict__ nxt, bool more1, bool more2, uint32_t kt)
                     __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int cb = ks & 1, nb = (ks + 1) & 1;
            if (ks == 1) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifndef W4_NOXFER
                if (more2) transfer(const_cast<uint32_t*>(cur), kt + 2u);
#endif
            }
            const uint32_t* src = ks == 1 ? nxt : cur;
            const int nks = ks == 1 ? 0 : 1;
            const bool rd = ks < 1 || more1;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
#pragma unroll
                for (int n = 0; n < 8; ++n)
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[m][n]) : "v"(fa[cb][m]), "v"(fb[cb][n]));
                __builtin_amdgcn_sched_barrier(0);
#ifndef W4_NORD
                if (rd) {
                    // two reads behind every eight MFMAs: b0 a0 | b1 b2 | b3 b4 | b5 b6 | b7 a1 | a2 a3 | a4 a5 | a6 a7
                    if (m == 0) { fb[nb][0] = rd_b(src, nks, 0); fa[nb][0] = rd_a(src, nks, 0); }
                    if (m == 1) { fb[nb][1] = rd_b(src, nks, 1); fb[nb][2] = rd_b(src, nks, 2); }
                    if (m == 2) { fb[nb][3] = rd_b(src, nks, 3); fb[nb][4] = rd_b(src, nks, 4); }
                    if (m == 3) { fb[nb][5] = rd_b(src, nks, 5); fb[nb][6] = rd_b(src, nks, 6); }
                    if (m == 4) { fb[nb][7] = rd_b(src, nks, 7); fa[nb][1] = rd_a(src, nks, 1); }
                    if (m == 5) { fa[nb][2] = rd_a(src, nks, 2); fa[nb][3] = rd_a(src, nks, 3); }
                    if (m == 6) { fa[nb][4] = rd_a(src, nks, 4); fa[nb][5] = rd_a(src, nks, 5); }
                    if (m == 7) { fa[nb][6] = rd_a(src, nks, 6); fa[nb][7] = rd_a(src, nks, 7); }
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    transfer(lds0, 0u);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (nk > 1u) transfer(lds1, 1u);
#pragma unroll
    for (int i = 0; i < 8; ++i) { fb[0][i] = rd_b(lds0, 0, i); fa[0][i] = rd_a(lds0, 0, i); }
    for (uint32_t kt = 0; kt < nk; kt += 2) {
        kstep(lds0, lds1, kt + 1u < nk, kt + 2u < nk, kt);
        if (kt + 1u < nk) kstep(lds1, lds0, kt + 2u < nk, kt + 3u < nk, kt + 1u);
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    if (tid == 0 && (blockIdx.x & 63u) == 7u) {
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        clk[2 * (blockIdx.x >> 6)] = c1 - c0;
        clk[2 * (blockIdx.x >> 6) + 1] = r1 - r0;
    }
    float s_ = 0.0f;
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 8; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) s_ += acc[m][n][r];
    out[(size_t)blockIdx.x * 256 + tid] = s_;
}

#ifdef W4_SHAPE16
#define w4_kernel w4_kernel16
#endif

int main(int argc, char** argv) {
    const uint32_t n_rows = argc > 1 ? (uint32_t)atol(argv[1]) : 1000000u;
    const uint32_t d = argc > 2 ? (uint32_t)atol(argv[2]) : 768u;
    const int reps = argc > 3 ? atoi(argv[3]) : 10;
    uint16_t *mirror, *qb;
    float *out, *thr;
    uint32_t* hits;
    const size_t mb = (size_t)n_rows * d * 2;
    CHECK(hipMalloc(&mirror, mb));
    CHECK(hipMalloc(&qb, (size_t)BM * d * 2));
    const uint32_t tiles = (n_rows + BN - 1) / BN;
    CHECK(hipMalloc(&out, (size_t)tiles * 256 * 4));
    CHECK(hipMalloc(&thr, BM * 4));
    CHECK(hipMalloc(&hits, 4));
    {
        // bf16 values in about [-1, 1): 0x3Fxx / 0xBFxx patterns with varying mantissas
        std::vector<uint16_t> h((size_t)1 << 22);
        uint32_t x = 12345u;
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (uint16_t)(((x >> 31) << 15) | 0x3C00u | ((x >> 8) & 0x3FFu)); }
        for (size_t off = 0; off < mb; off += h.size() * 2)
            CHECK(hipMemcpy((char*)mirror + off, h.data(), std::min(h.size() * 2, mb - off), hipMemcpyHostToDevice));
        CHECK(hipMemcpy(qb, h.data() + 777, (size_t)BM * d * 2, hipMemcpyHostToDevice));
        std::vector<float> t(BM, 1e30f);
        CHECK(hipMemcpy(thr, t.data(), BM * 4, hipMemcpyHostToDevice));
        CHECK(hipMemset(hits, 0, 4));
    }
    unsigned long long* clk;
    CHECK(hipMalloc(&clk, (size_t)(tiles / 64 + 1) * 16));
    CHECK(hipMemset(clk, 0, (size_t)(tiles / 64 + 1) * 16));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(w4_kernel, dim3(tiles), dim3(256), 0, 0, mirror, qb, n_rows, d, out, thr, hits, clk);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(w4_kernel, dim3(tiles), dim3(256), 0, 0, mirror, qb, n_rows, d, out, thr, hits, clk);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double flop = 2.0 * n_rows * d * BM;
    printf("w4_probe%s rows %u dim %u epi %d: %.1f us per pass, %.0f TFLOP/s, %.2f TB/s of mirror bytes\n", 
#ifdef W4_SHAPE16
           " 16x16x32",
#else
           " 32x32x16",
#endif
           n_rows, d, W4_EPI, ms * 1e3,
           flop / ms * 1e-9, (double)mb / ms * 1e-9);
    {
        std::vector<unsigned long long> h((size_t)(tiles / 64 + 1) * 2);
        CHECK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> ghz, cyc;
        for (size_t i = 0; i + 1 < h.size(); i += 2)
            if (h[i + 1]) { ghz.push_back((double)h[i] / (double)h[i + 1] * 0.1); cyc.push_back((double)h[i]); }
        if (!ghz.empty()) {
            std::sort(ghz.begin(), ghz.end());
            std::sort(cyc.begin(), cyc.end());
            const double mf = 16.0 * 4.0 * (d / 64) * 32.0;
            printf("  in-kernel clock (median of %zu workgroups): %.2f GHz; cycles per workgroup tile %.0f, of which MFMAs %.0f (%.0f %%)\n",
                   ghz.size(), ghz[ghz.size() / 2], cyc[cyc.size() / 2], mf, 100.0 * mf / cyc[cyc.size() / 2]);
        }
    }
    return 0;
}
