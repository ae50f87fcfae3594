// Reciprocal Rank Fusion for a batch of queries, on the device
// (FusionStrategy::reciprocal_rank_fusion, src/fusion.rs:68-84 of the reference, once per query).
//
// The dense top-k and the BM25 top-k of a query both come out of device-entry searches
// (trag_store_search_device, trag_bm25_search_device); fusing them here keeps the hybrid step of
// src/retrieve.rs:175-220 free of any host round trip, and B queries cost one launch. Ids are u32
// in ONE id space (row index = document index: every chunk went into both indices in the same
// order). One workgroup per query; the lists are at most FUSE_MAX entries each, so the quadratic
// forms below (find an id's first occurrence; rank by counting) are a few thousand LDS reads per
// thread at most and tens for the reference's 50 + 50 entries.
//
// Arithmetic and order are those of trag_fuse (csrc/trag_fuse.cpp): an id's score is the sum, in
// list order (dense list, then sparse list), of 1.0 / ((k + rank) + 1.0), each operation rounded
// once; results are ordered by score descending, equal scores by first appearance.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "trag.h"
#include "trag_error.h"

namespace trag {

constexpr uint32_t FUSE_MAX = 1024;   // entries per input list

__global__ void __launch_bounds__(256)
fuse_rrf_kernel(const uint32_t* __restrict__ dense_ids, const uint32_t* __restrict__ dense_n, uint32_t dense_stride,
                const uint32_t* __restrict__ sparse_ids, const uint32_t* __restrict__ sparse_n,
                uint32_t sparse_stride, float k, uint32_t k_out, uint32_t out_stride,
                uint32_t* __restrict__ out_ids, float* __restrict__ out_scores, uint32_t* __restrict__ out_n) {
    __shared__ uint32_t ids[2 * FUSE_MAX];
    __shared__ float score[2 * FUSE_MAX];   // of an id's first occurrence; NaN marks a later occurrence
    __shared__ uint32_t n_first;
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    const uint32_t nd = min(dense_n ? dense_n[b] : dense_stride, min(dense_stride, FUSE_MAX));
    const uint32_t ns = min(sparse_n ? sparse_n[b] : sparse_stride, min(sparse_stride, FUSE_MAX));
    const uint32_t n = nd + ns;
    for (uint32_t i = tid; i < n; i += 256u)
        ids[i] = i < nd ? dense_ids[(size_t)b * dense_stride + i] : sparse_ids[(size_t)b * sparse_stride + (i - nd)];
    if (tid == 0) n_first = 0u;
    __syncthreads();
    // entry i of the concatenated list: rank within its own list r(i); the score of an id is
    // accumulated by the thread that owns its first occurrence, in list order
    for (uint32_t i = tid; i < n; i += 256u) {
        const uint32_t id = ids[i];
        bool first = true;
        for (uint32_t j = 0; j < i; ++j)
            if (ids[j] == id) { first = false; break; }
        float s = __builtin_nanf("");
        if (first) {
            s = 0.0f;
            for (uint32_t j = i; j < n; ++j) {
                if (ids[j] != id) continue;
                float den = k + (float)(j < nd ? j : j - nd);
                den = den + 1.0f;
                s = s + 1.0f / den;
            }
            atomicAdd(&n_first, 1u);
        }
        score[i] = s;
    }
    __syncthreads();
    // position of a first occurrence = how many first occurrences come before it (score descending,
    // NaN last -- none can arise here, the inputs' scores are never read --, then first appearance)
    const uint32_t total = n_first;
    const uint32_t n_out = min(total, k_out);
    for (uint32_t i = tid; i < n; i += 256u) {
        const float s = score[i];
        if (s != s) continue;
        uint32_t pos = 0;
        for (uint32_t j = 0; j < n; ++j) {
            const float t = score[j];
            if (t != t) continue;
            pos += (t > s || (t == s && j < i)) ? 1u : 0u;
        }
        if (pos < n_out) {
            out_ids[(size_t)b * out_stride + pos] = ids[i];
            out_scores[(size_t)b * out_stride + pos] = s;
        }
    }
    if (tid == 0) out_n[b] = n_out;
}

}  // namespace trag

extern "C" trag_status trag_fuse_rrf_device(const uint32_t* d_dense_ids, const uint32_t* d_dense_n,
                                            uint32_t dense_stride, const uint32_t* d_sparse_ids,
                                            const uint32_t* d_sparse_n, uint32_t sparse_stride, uint32_t n_queries,
                                            float k, uint32_t k_out, uint32_t* d_out_ids, float* d_out_scores,
                                            uint32_t* d_out_n, void* hip_stream) {
    auto bad = [](const char* what) { return trag::set_error(TRAG_INVALID_ARG, std::string("trag_fuse_rrf_device: ") + what); };
    if (n_queries == 0) return TRAG_OK;
    if (!d_out_n || (k_out && (!d_out_ids || !d_out_scores))) return bad("null output array");
    if ((dense_stride && !d_dense_ids) || (sparse_stride && !d_sparse_ids)) return bad("a non-empty input list has no ids");
    if (dense_stride > trag::FUSE_MAX || sparse_stride > trag::FUSE_MAX) return bad("an input list is longer than 1024 entries");
    hipLaunchKernelGGL(trag::fuse_rrf_kernel, dim3(n_queries), dim3(256), 0, (hipStream_t)hip_stream, d_dense_ids,
                       d_dense_n, dense_stride, d_sparse_ids, d_sparse_n, sparse_stride, k, k_out, k_out, d_out_ids,
                       d_out_scores, d_out_n);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return trag::set_error(TRAG_DEVICE_ERROR, std::string("trag_fuse_rrf_device: ") + hipGetErrorString(e));
    return TRAG_OK;
}
