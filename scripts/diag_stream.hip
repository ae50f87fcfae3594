// Read-pattern probes for MI355X: how fast can a kernel shaped like the scans read HBM, and
// what does each of the scan's side outputs cost? Standalone (not part of libtrag.so):
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/diag_stream.hip -o trueno-rag_amd/lib/diag_stream
//   trueno-rag_amd/lib/diag_stream [rows]
// Every variant reads the same buffer of `rows` x 768 bytes (a bf16 mirror of rows x 384, default
// 10M rows = 7.68 GB) once per launch and prints the best of 5 launches. Seen on MI355X:
//   contiguous grid-stride, non-temporal loads (4 waves interleaved 1 KB apart)   6.3 TB/s
//   each wave reading its own 48 KB tile contiguously, non-temporal              7.03
//   the mirror scan's pattern (4 rows x 256 B per load), non-temporal            7.1
//   the same with plain (cached) loads                                           6.3
//   + bf16 -> f32, FMAs, 16-lane butterflies                                     7.1
//   + one 256-byte key store per tile: plain 6.3, non-temporal 6.65, system scope 6.65
//   + 64-byte store per tile 6.75, one dword per tile 6.92
//   + whole tile stored for every 4th tile 6.87, every 16th 7.02
//   + stores buffered in LDS and flushed 16 tiles at a time 6.35
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void eat(u32x4& acc, const u32x4 x) { acc ^= x; }

// P: contiguous grid-stride, U loads in flight per lane
template <int U>
__global__ void __launch_bounds__(256) p_contig(const u32x4* __restrict__ p, unsigned long long n16, uint32_t* sink) {
    u32x4 acc = {0, 0, 0, 0};
    const unsigned long long stride = (unsigned long long)gridDim.x * 256ull * U;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256ull * U + threadIdx.x; i < n16; i += stride) {
        u32x4 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned long long j = i + (unsigned long long)u * 256ull;
            x[u] = j < n16 ? __builtin_nontemporal_load(p + j) : acc;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) eat(acc, x[u]);
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;
}

// T: each wave owns 64-row tiles (48 KB contiguous, tile t = wave id + k * n_waves) and reads a
// tile with 48 fully contiguous wave-wide loads, U in flight, ping-pong groups
template <int U, bool NT>
__global__ void __launch_bounds__(256) p_tile_contig(const u32x4* __restrict__ p, uint32_t n_tiles, uint32_t* sink) {
    u32x4 acc = {0, 0, 0, 0};
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint32_t n_waves = gridDim.x * 4u;
    u32x4 xa[U], xb[U];
    auto issue = [&](u32x4 (&x)[U], uint32_t t, int g) {
        const u32x4* b = p + (size_t)t * 3072u + (size_t)g * U * 64 + lane;
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = NT ? __builtin_nontemporal_load(b + u * 64) : b[u * 64];
    };
    auto reduce = [&](const u32x4 (&x)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) eat(acc, x[u]);
    };
    constexpr int GROUPS = 48 / U;
    uint32_t t = blockIdx.x * 4u + wid;
    if (t < n_tiles) issue(xa, t, 0);
    for (; t < n_tiles; t += n_waves) {
#pragma unroll 1
        for (int g = 0; g < GROUPS; g += 2) {
            issue(xb, t, g + 1);
            reduce(xa);
            if (g + 2 < GROUPS) issue(xa, t, g + 2);
            else if (t + n_waves < n_tiles) issue(xa, t + n_waves, 0);
            reduce(xb);
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;
}

// M: the mirror scan's pattern for d = 384: 16 lanes per row, iteration `it` of a tile takes
// rows sub*16 + it (sub = lane / 16), 3 pieces per lane at +0, +256, +512 bytes; U iterations
// per group, ping-pong groups
template <int U, bool NT>
__global__ void __launch_bounds__(256) p_mirror(const u32x4* __restrict__ p, uint32_t n_tiles, uint32_t* sink) {
    u32x4 acc = {0, 0, 0, 0};
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int gl = lane & 15, sub = lane >> 4;
    const uint32_t n_waves = gridDim.x * 4u;
    u32x4 xa[U][3], xb[U][3];
    auto issue = [&](u32x4 (&x)[U][3], uint32_t t, int it0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const u32x4* b = p + ((size_t)t * 64u + (size_t)(sub * 16 + it0 + u)) * 48u + gl;
#pragma unroll
            for (int j = 0; j < 3; ++j) x[u][j] = NT ? __builtin_nontemporal_load(b + j * 16) : b[j * 16];
        }
    };
    auto reduce = [&](const u32x4 (&x)[U][3]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < 3; ++j) eat(acc, x[u][j]);
    };
    uint32_t t = blockIdx.x * 4u + wid;
    if (t < n_tiles) issue(xa, t, 0);
    for (; t < n_tiles; t += n_waves) {
#pragma unroll 1
        for (int it0 = 0; it0 < 16; it0 += 2 * U) {
            issue(xb, t, it0 + U);
            reduce(xa);
            if (it0 + 2 * U < 16) issue(xa, t, it0 + 2 * U);
            else if (t + n_waves < n_tiles) issue(xa, t + n_waves, 0);
            reduce(xb);
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;
}

// R: mirror pattern with consecutive rows per wave load: iteration `it` takes rows 4*it + sub
template <int U, bool NT>
__global__ void __launch_bounds__(256) p_mirror_rows(const u32x4* __restrict__ p, uint32_t n_tiles, uint32_t* sink) {
    u32x4 acc = {0, 0, 0, 0};
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int gl = lane & 15, sub = lane >> 4;
    const uint32_t n_waves = gridDim.x * 4u;
    u32x4 xa[U][3], xb[U][3];
    auto issue = [&](u32x4 (&x)[U][3], uint32_t t, int it0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const u32x4* b = p + ((size_t)t * 64u + (size_t)((it0 + u) * 4 + sub)) * 48u + gl;
#pragma unroll
            for (int j = 0; j < 3; ++j) x[u][j] = NT ? __builtin_nontemporal_load(b + j * 16) : b[j * 16];
        }
    };
    auto reduce = [&](const u32x4 (&x)[U][3]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < 3; ++j) eat(acc, x[u][j]);
    };
    uint32_t t = blockIdx.x * 4u + wid;
    if (t < n_tiles) issue(xa, t, 0);
    for (; t < n_tiles; t += n_waves) {
#pragma unroll 1
        for (int it0 = 0; it0 < 16; it0 += 2 * U) {
            issue(xb, t, it0 + U);
            reduce(xa);
            if (it0 + 2 * U < 16) issue(xa, t, it0 + 2 * U);
            else if (t + n_waves < n_tiles) issue(xa, t + n_waves, 0);
            reduce(xb);
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;
}


// A: the mirror pattern with the scan's arithmetic added in stages.
//   ARITH 1: bf16 -> f32 and the FMAs against a query held in registers; 2: + the 16-lane
//   butterfly per row; 3: + one 4-byte key store per row (coalesced, per tile)
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void piece(const v4f qlo, const v4f qhi, const u32x4 w, float& a0, float& a1) {
    a0 = __builtin_fmaf(qlo.x, __uint_as_float(w.x << 16), a0);
    a1 = __builtin_fmaf(qlo.y, __uint_as_float(w.x & 0xFFFF0000u), a1);
    a0 = __builtin_fmaf(qlo.z, __uint_as_float(w.y << 16), a0);
    a1 = __builtin_fmaf(qlo.w, __uint_as_float(w.y & 0xFFFF0000u), a1);
    a0 = __builtin_fmaf(qhi.x, __uint_as_float(w.z << 16), a0);
    a1 = __builtin_fmaf(qhi.y, __uint_as_float(w.z & 0xFFFF0000u), a1);
    a0 = __builtin_fmaf(qhi.z, __uint_as_float(w.w << 16), a0);
    a1 = __builtin_fmaf(qhi.w, __uint_as_float(w.w & 0xFFFF0000u), a1);
}
template <int U, int ARITH>
__global__ void __launch_bounds__(256) p_mirror_arith(const u32x4* __restrict__ p, uint32_t n_tiles, uint32_t* sink,
                                                      const float* __restrict__ q, float* __restrict__ keys) {
    __shared__ float lbuf[4][16][64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int gl = lane & 15, sub = lane >> 4;
    const uint32_t n_waves = gridDim.x * 4u;
    v4f qlo[3], qhi[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        qlo[j] = reinterpret_cast<const v4f*>(q)[2 * (j * 16 + gl)];
        qhi[j] = reinterpret_cast<const v4f*>(q)[2 * (j * 16 + gl) + 1];
    }
    u32x4 xa[U][3], xb[U][3];
    auto issue = [&](u32x4 (&x)[U][3], uint32_t t, int it0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const u32x4* b = p + ((size_t)t * 64u + (size_t)(sub * 16 + it0 + u)) * 48u + gl;
#pragma unroll
            for (int j = 0; j < 3; ++j) x[u][j] = __builtin_nontemporal_load(b + j * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    float mine = 0.0f, total = 0.0f;
    auto reduce = [&](const u32x4 (&x)[U][3], int it0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
            for (int j = 0; j < 3; ++j) piece(qlo[j], qhi[j], x[u][j], a0, a1);
            a0 += a1;
            if (ARITH >= 2) {
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) a0 += __shfl_xor(a0, o);
            }
            if (it0 + u == gl) mine = a0;
            total += a0;
        }
    };
    uint32_t t = blockIdx.x * 4u + wid;
    if (t < n_tiles) issue(xa, t, 0);
    for (; t < n_tiles; t += n_waves) {
#pragma unroll 1
        for (int it0 = 0; it0 < 16; it0 += 2 * U) {
            issue(xb, t, it0 + U);
            reduce(xa, it0);
            const bool same = it0 + 2 * U < 16;
            const uint32_t tn = same ? t : (t + n_waves < n_tiles ? t + n_waves : t);
            issue(xa, tn, same ? it0 + 2 * U : 0);
            reduce(xb, it0 + U);
        }
        if (ARITH == 3) keys[(size_t)t * 64u + lane] = mine;
        if (ARITH == 8 && ((t / n_waves) & 3u) == 0u) keys[(size_t)t * 64u + lane] = mine;    // every 4th tile, whole
        if (ARITH == 9 && ((t / n_waves) & 15u) == 0u) keys[(size_t)t * 64u + lane] = mine;   // every 16th tile
        if (ARITH == 10) {   // buffered in LDS, flushed 16 tiles at a time
            const uint32_t slot = (t / n_waves) & 15u;
            lbuf[wid][slot][lane] = mine;
            if (slot == 15u) {
#pragma unroll
                for (uint32_t s2 = 0; s2 < 16u; ++s2)
                    keys[(size_t)(t - (15u - s2) * n_waves) * 64u + lane] = lbuf[wid][s2][lane];
            }
        }
        if (ARITH == 4) __builtin_nontemporal_store(mine, keys + (size_t)t * 64u + lane);
        if (ARITH == 5 && lane < 16) keys[(size_t)t * 16u + lane] = mine;          // a quarter of the stores
        if (ARITH == 6 && lane == 0) keys[t] = mine;                                // one dword per tile
        if (ARITH == 7) __hip_atomic_store(keys + (size_t)t * 64u + lane, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (total + mine == 12345.678f) sink[0] = 1u;
}

template <typename F>
static void run(const char* name, double bytes, F launch) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 0.0f;
    for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(e0, 0));
        launch();
        CHECK(hipGetLastError());
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.0f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (best == 0.0f || ms < best) best = ms;
    }
    printf("%-44s %8.3f ms  %8.1f GB/s\n", name, best, bytes / (best * 1e-3) / 1e9);
    fflush(stdout);
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
    const uint64_t rows = argc > 1 ? strtoull(argv[1], nullptr, 10) : 10000000ull;
    const uint64_t n_tiles64 = rows / 64;           // whole tiles only
    const uint64_t bytes = n_tiles64 * 64 * 768;
    const unsigned long long n16 = bytes / 16;
    const uint32_t n_tiles = (uint32_t)n_tiles64;
    u32x4* buf = nullptr;
    uint32_t* sink = nullptr;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&sink, 256));
    CHECK(hipMemset(buf, 1, bytes));
    CHECK(hipDeviceSynchronize());
    const double B = (double)bytes;
    printf("buffer %.2f GB\n", B / 1e9);
#define L(K, G) [&] { hipLaunchKernelGGL(K, dim3(G), dim3(256), 0, 0, buf, n16, sink); }
#define LT(K, G) [&] { hipLaunchKernelGGL(K, dim3(G), dim3(256), 0, 0, buf, n_tiles, sink); }
    run("contig U=8  grid 2048", B, L(p_contig<8>, 2048));
    run("contig U=8  grid 1024", B, L(p_contig<8>, 1024));
    run("contig U=8  grid 512", B, L(p_contig<8>, 512));
    run("contig U=8  grid 256", B, L(p_contig<8>, 256));
    run("contig U=16 grid 256", B, L(p_contig<16>, 256));
    run("contig U=24 grid 256", B, L(p_contig<24>, 256));
    run("contig U=16 grid 512", B, L(p_contig<16>, 512));
    run("tile-contig U=12x2 nt grid 256", B, LT((p_tile_contig<12, true>), 256));
    run("tile-contig U=12x2 nt grid 512", B, LT((p_tile_contig<12, true>), 512));
    run("tile-contig U=6x2 nt grid 256", B, LT((p_tile_contig<6, true>), 256));
    run("tile-contig U=6x2 nt grid 512", B, LT((p_tile_contig<6, true>), 512));
    run("tile-contig U=6x2 nt grid 1024", B, LT((p_tile_contig<6, true>), 1024));
    run("tile-contig U=12x2 plain grid 256", B, LT((p_tile_contig<12, false>), 256));
    run("mirror U=4x2 nt grid 256", B, LT((p_mirror<4, true>), 256));
    run("mirror U=4x2 nt grid 512", B, LT((p_mirror<4, true>), 512));
    run("mirror U=2x2 nt grid 256", B, LT((p_mirror<2, true>), 256));
    run("mirror U=2x2 nt grid 512", B, LT((p_mirror<2, true>), 512));
    run("mirror U=2x2 nt grid 1024", B, LT((p_mirror<2, true>), 1024));
    run("mirror U=4x2 plain grid 256", B, LT((p_mirror<4, false>), 256));
    run("mirror-rows U=4x2 nt grid 256", B, LT((p_mirror_rows<4, true>), 256));
    run("mirror-rows U=2x2 nt grid 512", B, LT((p_mirror_rows<2, true>), 512));
    run("mirror-rows U=4x2 plain grid 256", B, LT((p_mirror_rows<4, false>), 256));
    float *q = nullptr, *keys = nullptr;
    CHECK(hipMalloc(&q, 384 * 4));
    CHECK(hipMemset(q, 0, 384 * 4));
    CHECK(hipMalloc(&keys, (size_t)n_tiles * 64 * 4));
#define LA(K, G) [&] { hipLaunchKernelGGL(K, dim3(G), dim3(256), 0, 0, buf, n_tiles, sink, q, keys); }
    run("mirror U=4x2 + fma                grid 256", B, LA((p_mirror_arith<4, 1>), 256));
    run("mirror U=4x2 + fma + butterfly    grid 256", B, LA((p_mirror_arith<4, 2>), 256));
    run("mirror U=4x2 + fma + bfly + keys  grid 256", B, LA((p_mirror_arith<4, 3>), 256));
    run("mirror U=4x2 ... + keys nt store  grid 256", B, LA((p_mirror_arith<4, 4>), 256));
    run("mirror U=4x2 ... + 16-lane store  grid 256", B, LA((p_mirror_arith<4, 5>), 256));
    run("mirror U=4x2 ... + 1 dword / tile grid 256", B, LA((p_mirror_arith<4, 6>), 256));
    run("mirror U=4x2 ... + system-scope st grid 256", B, LA((p_mirror_arith<4, 7>), 256));
    run("mirror U=4x2 ... + every 4th tile  grid 256", B, LA((p_mirror_arith<4, 8>), 256));
    run("mirror U=4x2 ... + every 16th tile grid 256", B, LA((p_mirror_arith<4, 9>), 256));
    run("mirror U=4x2 ... + LDS-batched x16 grid 256", B, LA((p_mirror_arith<4, 10>), 256));
    CHECK(hipFree(buf));
    CHECK(hipFree(sink));
    return 0;
}
