// EXPERIMENT, NOT PART OF THE LIBRARY: can a batch pass over f32 rows be served by converting the corpus to bf16 chunk by
// chunk into a small buffer that is re-read at once (the direct-to-LDS GEMM of the mirror form running on the chunk while
// the next chunk is converted), so that the bf16 bytes never travel to HBM and back? Timed here: the conversion alone into
// a full-size destination, into two small buffers used in turn, and the latter with a reader of the chunk just written on a
// second stream (a plain streaming read standing in for the GEMM's operand traffic).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/experiments/chunk_convert_probe.hip -o <out>; <out> [rows] [dim] [chunk rows]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(256) cvt_kernel(const v4f* __restrict__ src, u32x4* __restrict__ dst, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const v4f a = __builtin_nontemporal_load(&src[2 * i]), b = __builtin_nontemporal_load(&src[2 * i + 1]);
        const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_convertvector(a, bf16x4));
        const u32x2 hi = __builtin_bit_cast(u32x2, __builtin_convertvector(b, bf16x4));
        dst[i] = u32x4{lo.x, lo.y, hi.x, hi.y};
    }
}

__global__ void __launch_bounds__(256) read_kernel(const u32x4* __restrict__ src, size_t n, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const u32x4 v = src[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
    const size_t rows = argc > 1 ? (size_t)atol(argv[1]) : 1000000;
    const size_t d = argc > 2 ? (size_t)atol(argv[2]) : 768;
    const size_t crow = argc > 3 ? (size_t)atol(argv[3]) : 65536;
    float* src;
    uint16_t *full, *buf[2];
    uint32_t* sink;
    CHECK(hipMalloc(&src, rows * d * 4));
    CHECK(hipMalloc(&full, rows * d * 2));
    CHECK(hipMalloc(&buf[0], crow * d * 2));
    CHECK(hipMalloc(&buf[1], crow * d * 2));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(src, 0x3c, rows * d * 4));
    hipStream_t a, b;
    CHECK(hipStreamCreate(&a));
    CHECK(hipStreamCreate(&b));
    hipEvent_t e0, e1, done[2], read_done[2];
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) { CHECK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming)); CHECK(hipEventCreateWithFlags(&read_done[i], hipEventDisableTiming)); }
    const int grid = 256 * 8;
    auto timeit = [&](const char* name, auto&& body) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipDeviceSynchronize();
            hipEventRecord(e0, a);
            body();
            hipEventRecord(e1, a);
            hipDeviceSynchronize();
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-70s %.3f ms  (%.2f TB/s of f32 bytes)\n", name, best, rows * d * 4 / best * 1e-9);
    };
    timeit("conversion, full-size destination", [&] {
        hipLaunchKernelGGL(cvt_kernel, dim3(grid), dim3(256), 0, a, (const v4f*)src, (u32x4*)full, rows * d / 8);
    });
    timeit("conversion in chunks, full-size destination", [&] {
        for (size_t r = 0; r < rows; r += crow) {
            const size_t n = (rows - r < crow ? rows - r : crow) * d;
            hipLaunchKernelGGL(cvt_kernel, dim3(grid), dim3(256), 0, a, (const v4f*)(src + r * d), (u32x4*)(full + r * d), n / 8);
        }
    });
    timeit("conversion in chunks, two small destinations in turn", [&] {
        int x = 0;
        for (size_t r = 0; r < rows; r += crow, x ^= 1) {
            const size_t n = (rows - r < crow ? rows - r : crow) * d;
            hipLaunchKernelGGL(cvt_kernel, dim3(grid), dim3(256), 0, a, (const v4f*)(src + r * d), (u32x4*)buf[x], n / 8);
        }
    });
    timeit("... and every chunk re-read at once on a second stream", [&] {
        int x = 0;
        size_t c = 0;
        for (size_t r = 0; r < rows; r += crow, x ^= 1, ++c) {
            const size_t n = (rows - r < crow ? rows - r : crow) * d;
            if (c >= 2) hipStreamWaitEvent(a, read_done[x], 0);
            hipLaunchKernelGGL(cvt_kernel, dim3(grid), dim3(256), 0, a, (const v4f*)(src + r * d), (u32x4*)buf[x], n / 8);
            hipEventRecord(done[x], a);
            hipStreamWaitEvent(b, done[x], 0);
            hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, b, (const u32x4*)buf[x], n / 8, sink);
            hipEventRecord(read_done[x], b);
        }
        hipStreamWaitEvent(a, read_done[0], 0);
        hipStreamWaitEvent(a, read_done[1], 0);
    });
    timeit("plain read of the f32 rows (the pass's floor)", [&] {
        hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, a, (const u32x4*)src, rows * d / 4, sink);
    });
    return 0;
}
