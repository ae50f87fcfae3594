// BM25 sparse index of the C ABI (include/trag.h, trag_bm25_*): the sparse half
// of the hybrid retrieve (src/retrieve.rs:183-187 of the reference), replacing
// BM25Index (src/index.rs:31-280).
//
// Layout. The host keeps each document's distinct terms and frequencies
// (forward lists, needed to undo a document on removal) and the per-term
// document frequencies; HBM holds the inverted lists in CSR form, one 8-byte
// posting {document, term frequency} per (term, document), documents ascending
// within a term, plus one u32 length and one f32 accumulator per document.
//
// A query is scored with one launch per query term, in query order: the launch
// walks that term's postings (coalesced 8-byte loads) and adds idf * tf_norm to
// the documents' accumulators. Launch order on one stream is what reproduces
// the reference's per-document f32 sum over the query's terms
// (src/index.rs:230-233): a document appears at most once per posting list, so
// within a launch no two lanes touch the same accumulator. A last pass turns
// accumulators into ordered score keys (score <= 0 -> not a result,
// src/index.rs:236), clears them, and feeds the dense path's select.
//
// This file is compiled with -ffp-contract=off: every operation of the score
// formula rounds once, as in the reference (src/index.rs:146-153).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "trag.h"
#include "trag_device.h"
#include "trag_internal.h"
#include "trag_scan_common.h"

using namespace trag;

namespace trag {

// acc[doc] += idf * tf_norm(tf, len[doc]) for every posting of one term.
__global__ void __launch_bounds__(256)
bm25_term_kernel(const uint2* __restrict__ post, uint32_t count, const uint32_t* __restrict__ doc_len,
                 float* __restrict__ acc, float idf, float k1, float k1_plus_1, float one_minus_b,
                 float b, float avg_len) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const uint2 p = post[i];
        const float tf = (float)p.y;
        const float len = (float)doc_len[p.x];
        // (tf * (k1 + 1)) / (tf + k1 * ((1 - b) + (b * len) / avg)), src/index.rs:150-151
        const float t1 = b * len;
        const float t2 = t1 / avg_len;
        const float t3 = one_minus_b + t2;
        const float t4 = k1 * t3;
        const float den = tf + t4;
        const float num = tf * k1_plus_1;
        const float tf_norm = num / den;
        const float s = idf * tf_norm;
        acc[p.x] = acc[p.x] + s;
    }
}

// Accumulators -> ordered keys, with the per-tile / per-wave maxima and the
// coarse histogram the select stages expect (tile t belongs to wave
// t % n_waves). Touched accumulators are cleared for the next query.
__global__ void __launch_bounds__(256)
bm25_keys_kernel(float* __restrict__ acc, uint32_t n_docs, uint32_t need, uint32_t k,
                 SearchCtl* __restrict__ ctl, uint32_t* __restrict__ keys, uint32_t* __restrict__ tmax,
                 uint32_t* __restrict__ wmax, uint32_t* __restrict__ hist) {
    __shared__ uint32_t lhist[HIST_BINS];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < (int)HIST_BINS; i += 256) lhist[i] = 0u;
    const uint32_t n_tiles = (n_docs + 63u) >> 6;
    const uint32_t n_waves = gridDim.x * 4u;
    const uint32_t wave_id = blockIdx.x * 4u + (uint32_t)wid;
    if (blockIdx.x == 0 && tid == 0) publish_ctl(ctl, 0.0f, need, k, nullptr, n_waves);
    __syncthreads();
    uint32_t wave_best = 0u;
    for (uint32_t t = wave_id; t < n_tiles; t += n_waves) {
        const uint32_t doc = t * 64u + (uint32_t)lane;
        uint32_t key = 0u;
        if (doc < n_docs) {
            const float a = acc[doc];
            if (a > 0.0f) {
                key = score_key(a);
                atomicAdd(&lhist[key >> 21], 1u);
            }
            if (a != 0.0f) acc[doc] = 0.0f;
            keys[doc] = key;
        }
        const uint32_t tm = wave_max_u32(key);
        if (lane == 0) tmax[t] = tm;
        wave_best = tm > wave_best ? tm : wave_best;
    }
    if (lane == 0) wmax[wave_id] = wave_best;
    __syncthreads();
    for (int i = tid; i < (int)HIST_BINS; i += 256) {
        const uint32_t h = lhist[i];
        if (h) atomicAdd(&hist[i], h);
    }
}

}  // namespace trag

namespace {

#define BM_TRY(expr)                                                                           \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            return set_error(_e == hipErrorOutOfMemory ? TRAG_OOM : TRAG_DEVICE_ERROR,         \
                             std::string(#expr) + ": " + hipGetErrorString(_e));               \
        }                                                                                      \
    } while (0)

struct DeviceScope {
    int prev = -1;
    bool switched = false;
    explicit DeviceScope(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceScope() {
        if (switched) (void)hipSetDevice(prev);
    }
};

constexpr uint32_t MAX_TERM_ID = 1u << 28;

}  // namespace

struct trag_bm25 {
    trag_bm25_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
    // host state
    std::vector<uint64_t> fwd_off{0};   // per document: range in fwd
    std::vector<uint2> fwd;             // {term, tf}, terms ascending within a document
    std::vector<uint32_t> doc_len;      // tokens per document
    std::vector<uint8_t> removed;
    std::vector<uint32_t> df;           // live documents per term
    uint64_t n_live = 0;
    uint64_t total_len = 0;             // tokens over live documents
    uint64_t live_postings = 0;
    std::vector<uint64_t> term_off;     // CSR offsets of the committed inverted lists
    bool dirty = true;
    bool full_select = false;           // learned: scores crowd one coarse bin
    // device state
    uint2* d_post = nullptr;
    uint64_t post_cap = 0;
    uint32_t* d_doc_len = nullptr;
    float* d_acc = nullptr;
    uint64_t doc_cap = 0;
    uint64_t n_docs_dev = 0;            // documents covered by the committed arrays
    Scratch scratch{};
    void* zero_block = nullptr;
    size_t zero_bytes = 0;
    uint32_t* d_out = nullptr;          // staging, laid out per call: k docs | k scores | n | status
    uint32_t* h_out = nullptr;          // its pinned host copy (one transfer per search)
    uint32_t out_cap_k = 0;
    // one scratch, one accumulator array: a call on another stream than the call before it first
    // waits for that call's work (as trag_store does)
    hipStream_t last_stream = nullptr;
    bool has_last = false;
    hipEvent_t order_ev = nullptr;
};

namespace {

void free_device(trag_bm25* ix) {
    void* ptrs[] = {ix->d_post, ix->d_doc_len, ix->d_acc, ix->scratch.keys, ix->scratch.tmax,
                    ix->scratch.wmax, ix->scratch.cand, ix->zero_block, ix->d_out};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    ix->d_post = nullptr;
    ix->d_doc_len = nullptr;
    ix->d_acc = nullptr;
    ix->scratch = Scratch{};
    ix->zero_block = nullptr;
    ix->d_out = nullptr;
    if (ix->h_out) (void)hipHostFree(ix->h_out);
    ix->h_out = nullptr;
    ix->post_cap = ix->doc_cap = 0;
    ix->out_cap_k = 0;
}

// Rebuild the inverted lists from the live documents' forward lists and upload them.
trag_status commit(trag_bm25* ix) {
    const uint64_t n_docs = ix->doc_len.size();
    const size_t n_terms = ix->df.size();
    ix->term_off.assign(n_terms + 1, 0);
    for (size_t t = 0; t < n_terms; ++t) ix->term_off[t + 1] = ix->term_off[t] + ix->df[t];
    const uint64_t nnz = ix->term_off[n_terms];
    std::vector<uint2> post(nnz);
    {
        std::vector<uint64_t> cur(ix->term_off.begin(), ix->term_off.end() - 1);
        for (uint64_t d = 0; d < n_docs; ++d) {
            if (ix->removed[d]) continue;
            for (uint64_t p = ix->fwd_off[d]; p < ix->fwd_off[d + 1]; ++p) {
                const uint2 e = ix->fwd[p];
                post[cur[e.x]++] = make_uint2((uint32_t)d, e.y);
            }
        }
    }
    hipStream_t st = ix->stream;
    // buffers may be replaced below: nothing enqueued earlier (device-entry searches on a
    // caller's stream included) may still be using them
    if (ix->d_post || ix->zero_block) BM_TRY(hipDeviceSynchronize());
    if (nnz > ix->post_cap) {
        if (ix->d_post) (void)hipFree(ix->d_post);
        ix->d_post = nullptr;
        ix->post_cap = 0;
        const uint64_t cap = std::max<uint64_t>(nnz, 1024);
        BM_TRY(hipMalloc(&ix->d_post, cap * sizeof(uint2)));
        ix->post_cap = cap;
    }
    if (n_docs > ix->doc_cap || !ix->zero_block) {
        void* old[] = {ix->d_doc_len, ix->d_acc, ix->scratch.keys, ix->scratch.tmax};
        for (void* p : old)
            if (p) (void)hipFree(p);
        ix->d_doc_len = nullptr;
        ix->d_acc = nullptr;
        ix->scratch.keys = nullptr;
        ix->scratch.tmax = nullptr;
        ix->doc_cap = 0;
        uint64_t cap = std::max<uint64_t>(n_docs + n_docs / 2, 1024);
        cap = (cap + 63) & ~63ull;
        BM_TRY(hipMalloc(&ix->d_doc_len, cap * sizeof(uint32_t)));
        BM_TRY(hipMalloc(&ix->d_acc, cap * sizeof(float)));
        BM_TRY(hipMalloc(&ix->scratch.keys, cap * sizeof(uint32_t)));
        BM_TRY(hipMalloc(&ix->scratch.tmax, (cap / 64 + 1) * sizeof(uint32_t)));
        BM_TRY(hipMemsetAsync(ix->d_acc, 0, cap * sizeof(float), st));
        ix->doc_cap = cap;
        if (!ix->zero_block) {
            const size_t ctl_bytes = 256;
            static_assert(sizeof(SearchCtl) <= 256, "control block grew");
            ix->zero_bytes = ctl_bytes + HIST_LEVELS * HIST_BINS * sizeof(uint32_t);
            BM_TRY(hipMalloc(&ix->zero_block, ix->zero_bytes));
            BM_TRY(hipMemsetAsync(ix->zero_block, 0, ix->zero_bytes, st));
            ix->scratch.ctl = reinterpret_cast<SearchCtl*>(ix->zero_block);
            ix->scratch.hist = reinterpret_cast<uint32_t*>(static_cast<char*>(ix->zero_block) + ctl_bytes);
            BM_TRY(hipMalloc(&ix->scratch.wmax, WMAX_CAP * sizeof(uint32_t)));
            BM_TRY(hipMalloc(&ix->scratch.cand, CAND_CAP * sizeof(uint32_t)));
        }
    }
    if (nnz) BM_TRY(hipMemcpyAsync(ix->d_post, post.data(), nnz * sizeof(uint2), hipMemcpyHostToDevice, st));
    if (n_docs)
        BM_TRY(hipMemcpyAsync(ix->d_doc_len, ix->doc_len.data(), n_docs * sizeof(uint32_t),
                              hipMemcpyHostToDevice, st));
    BM_TRY(hipStreamSynchronize(st));  // `post` is pageable host memory that dies here
    ix->n_docs_dev = n_docs;
    ix->dirty = false;
    return TRAG_OK;
}

trag_status ensure_out(trag_bm25* ix, uint32_t k) {
    if (k <= ix->out_cap_k && ix->d_out && ix->h_out) return TRAG_OK;
    if (ix->d_out) (void)hipFree(ix->d_out);
    if (ix->h_out) (void)hipHostFree(ix->h_out);
    ix->d_out = nullptr;
    ix->h_out = nullptr;
    ix->out_cap_k = 0;
    const uint32_t cap = std::max<uint32_t>(k, 64);
    BM_TRY(hipMalloc(&ix->d_out, ((size_t)2 * cap + 2) * sizeof(uint32_t)));
    BM_TRY(hipHostMalloc(reinterpret_cast<void**>(&ix->h_out), ((size_t)2 * cap + 2) * sizeof(uint32_t),
                         hipHostMallocDefault));
    ix->out_cap_k = cap;
    return TRAG_OK;
}

// Enqueue one query on st. Outputs are device pointers.
trag_status enqueue_query(trag_bm25* ix, const uint32_t* q_terms, uint32_t n_terms, uint32_t k,
                          uint32_t* d_docs, float* d_scores, uint32_t* d_n, uint32_t* d_status,
                          hipStream_t st) {
    if (ix->has_last && ix->last_stream != st && ix->order_ev) {
        hipError_t e = hipEventRecord(ix->order_ev, ix->last_stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(st, ix->order_ev, 0);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            BM_TRY(hipDeviceSynchronize());
        }
    }
    ix->last_stream = st;
    ix->has_last = true;
    if (ix->dirty) {
        trag_status ts = commit(ix);   // (uploads on the index's own stream and waits for them)
        if (ts != TRAG_OK) return ts;
    }
    const uint32_t n_docs = (uint32_t)ix->n_docs_dev;
    const uint32_t live = (uint32_t)ix->n_live;
    const uint32_t k_eff = std::min<uint32_t>(k, live);
    if (k_eff > CAND_CAP)
        return set_error(TRAG_UNSUPPORTED, "k exceeds the single-workgroup final sort (k <= " +
                                               std::to_string(CAND_CAP) + ")");
    // src/index.rs:143-147; ln through the C library, as Rust's f32::ln does
    const float n = (float)live;
    const float avg = (float)(uint32_t)ix->total_len / (float)live;  // src/index.rs:161-162
    const float k1 = ix->cfg.k1, b = ix->cfg.b;
    const float k1p1 = k1 + 1.0f, omb = 1.0f - b;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        n_cu = 256;
        if (hipGetDevice(&dev) == hipSuccess)
            (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    }
    for (uint32_t t = 0; t < n_terms; ++t) {
        const uint32_t term = q_terms[t];
        if (term >= ix->df.size()) continue;
        const uint64_t off = ix->term_off[term];
        const uint64_t cnt = ix->term_off[term + 1] - off;
        if (cnt == 0) continue;
        const float df = (float)(uint32_t)cnt;
        const float idf = logf((n - df + 0.5f) / (df + 0.5f) + 1.0f);
        const uint32_t grid = (uint32_t)std::min<uint64_t>((cnt + 255) / 256, (uint64_t)n_cu * 8);
        hipLaunchKernelGGL(bm25_term_kernel, dim3(grid), dim3(256), 0, st, ix->d_post + off, (uint32_t)cnt,
                           ix->d_doc_len, ix->d_acc, idf, k1, k1p1, omb, b, avg);
    }
    const uint32_t n_tiles = (n_docs + 63u) / 64u;
    const uint32_t kgrid = (uint32_t)std::min<uint64_t>(std::max<uint32_t>((n_tiles + 3) / 4, 1u),
                                                       std::min<uint64_t>((uint64_t)n_cu * 8, WMAX_CAP / 4));
    hipLaunchKernelGGL(bm25_keys_kernel, dim3(kgrid), dim3(256), 0, st, ix->d_acc, n_docs, k_eff, k_eff,
                       ix->scratch.ctl, ix->scratch.keys, ix->scratch.tmax, ix->scratch.wmax,
                       ix->scratch.hist);
    BM_TRY(hipGetLastError());
    const ScanArgs a{nullptr, nullptr, n_docs, 0u, (uint32_t)TRAG_DOT_PRODUCT};
    if (!ix->full_select) {
        BM_TRY(launch_select_final(nullptr, a, ix->scratch, false, live, k_eff, k_eff, kgrid * 4u, d_docs,
                                   d_scores, d_n, d_status, st));
    } else {
        BM_TRY(launch_select(a, ix->scratch, st));
        BM_TRY(launch_final(nullptr, a, ix->scratch, false, live, d_docs, d_scores, d_n, d_status, st));
        BM_TRY(hipMemsetAsync(ix->scratch.hist, 0, ix->zero_bytes - 256, st));
        BM_TRY(launch_reset_ctl(ix->scratch, st));
    }
    return TRAG_OK;
}

// One document's tokens -> its {term, tf} list (terms ascending).
void count_terms(const uint32_t* tok, uint64_t len, std::vector<uint32_t>& tmp, std::vector<uint2>& out) {
    tmp.assign(tok, tok + len);
    std::sort(tmp.begin(), tmp.end());
    for (size_t i = 0; i < tmp.size();) {
        size_t j = i + 1;
        while (j < tmp.size() && tmp[j] == tmp[i]) ++j;
        out.push_back(make_uint2(tmp[i], (uint32_t)(j - i)));
        i = j;
    }
}

trag_status add_docs(trag_bm25* ix, const uint32_t* terms, const uint64_t* doc_offsets, uint64_t n_docs,
                     uint32_t* out_first_doc) {
    const uint64_t first = ix->doc_len.size();
    if (first + n_docs > 0xFFFFFFF0ull) return set_error(TRAG_INVALID_ARG, "document count exceeds u32 ids");
    uint32_t max_term = 0;
    for (uint64_t i = 0; i < n_docs; ++i) {
        if (doc_offsets[i + 1] < doc_offsets[i]) return set_error(TRAG_INVALID_ARG, "doc_offsets must not decrease");
        if (doc_offsets[i + 1] - doc_offsets[i] > 0xFFFFFFFFull)
            return set_error(TRAG_INVALID_ARG, "document longer than u32 tokens");
    }
    for (uint64_t p = doc_offsets[0]; p < doc_offsets[n_docs]; ++p) max_term = std::max(max_term, terms[p]);
    if (max_term >= MAX_TERM_ID)
        return set_error(TRAG_INVALID_ARG, "term ids must be dense (below 2^28): map terms through a dictionary");
    if (doc_offsets[n_docs] > doc_offsets[0] && ix->df.size() <= max_term) ix->df.resize((size_t)max_term + 1, 0u);
    std::vector<uint32_t> tmp;
    for (uint64_t i = 0; i < n_docs; ++i) {
        const uint64_t len = doc_offsets[i + 1] - doc_offsets[i];
        const size_t before = ix->fwd.size();
        count_terms(terms + doc_offsets[i], len, tmp, ix->fwd);
        for (size_t p = before; p < ix->fwd.size(); ++p) ix->df[ix->fwd[p].x] += 1;
        ix->live_postings += ix->fwd.size() - before;
        ix->fwd_off.push_back(ix->fwd.size());
        ix->doc_len.push_back((uint32_t)len);
        ix->removed.push_back(0);
        ix->total_len += len;
        ix->n_live += 1;
    }
    if (out_first_doc) *out_first_doc = (uint32_t)first;
    ix->dirty = true;
    return TRAG_OK;
}

inline uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

}  // namespace

extern "C" {

trag_status trag_bm25_create(const trag_bm25_config* cfg, trag_bm25** out) {
    if (!cfg || !out) return set_error(TRAG_INVALID_ARG, "null argument");
    *out = nullptr;
    if (!(cfg->k1 >= 0.0f) || !(cfg->b >= 0.0f) || !(cfg->b <= 1.0f))
        return set_error(TRAG_INVALID_CONFIG, "k1 must be >= 0 and b within [0, 1]");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0)
        return set_error(TRAG_DEVICE_ERROR, "no HIP device visible (there is no CPU fallback)");
    int dev = cfg->device;
    if (dev < 0) BM_TRY(hipGetDevice(&dev));
    if (dev >= n_dev) return set_error(TRAG_INVALID_CONFIG, "device ordinal out of range");
    trag_bm25* ix = new trag_bm25();
    ix->cfg = *cfg;
    ix->device = dev;
    DeviceScope guard(dev);
    hipError_t e = hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ix->order_ev, hipEventDisableTiming);
    if (e != hipSuccess) {
        if (ix->stream) (void)hipStreamDestroy(ix->stream);
        delete ix;
        return set_error(TRAG_DEVICE_ERROR, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    *out = ix;
    return TRAG_OK;
}

void trag_bm25_destroy(trag_bm25* ix) {
    if (!ix) return;
    {
        DeviceScope guard(ix->device);
        if (ix->stream) (void)hipStreamSynchronize(ix->stream);
        free_device(ix);
        if (ix->order_ev) (void)hipEventDestroy(ix->order_ev);
        if (ix->stream) (void)hipStreamDestroy(ix->stream);
    }
    delete ix;
}

trag_status trag_bm25_add(trag_bm25* ix, const uint32_t* terms, const uint64_t* doc_offsets, uint64_t n_docs,
                          uint32_t* out_first_doc) {
    if (!ix || !doc_offsets) return set_error(TRAG_INVALID_ARG, "null argument");
    if (n_docs && doc_offsets[n_docs] > doc_offsets[0] && !terms) return set_error(TRAG_INVALID_ARG, "null terms");
    std::lock_guard<std::mutex> lk(ix->mu);
    return add_docs(ix, terms, doc_offsets, n_docs, out_first_doc);
}

trag_status trag_bm25_remove(trag_bm25* ix, uint32_t doc) {
    if (!ix) return set_error(TRAG_INVALID_ARG, "null index");
    std::lock_guard<std::mutex> lk(ix->mu);
    if (doc >= ix->doc_len.size() || ix->removed[doc]) return TRAG_OK;
    ix->removed[doc] = 1;
    for (uint64_t p = ix->fwd_off[doc]; p < ix->fwd_off[doc + 1]; ++p) ix->df[ix->fwd[p].x] -= 1;
    ix->live_postings -= ix->fwd_off[doc + 1] - ix->fwd_off[doc];
    ix->total_len -= ix->doc_len[doc];
    ix->n_live -= 1;
    ix->dirty = true;
    return TRAG_OK;
}

uint64_t trag_bm25_len(const trag_bm25* ix) { return ix ? ix->n_live : 0; }

trag_status trag_bm25_search_device(trag_bm25* ix, const uint32_t* query_terms, uint32_t n_terms, uint32_t k,
                                    uint32_t* d_out_docs, float* d_out_scores, uint32_t* d_out_n,
                                    uint32_t* d_status, void* hip_stream) {
    if (!ix) return set_error(TRAG_INVALID_ARG, "null index");
    if (!d_out_n || (n_terms && !query_terms) || (k && (!d_out_docs || !d_out_scores)))
        return set_error(TRAG_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceScope guard(ix->device);
    hipStream_t st = (hipStream_t)hip_stream;
    if (k == 0 || n_terms == 0 || ix->n_live == 0) {
        BM_TRY(hipMemsetAsync(d_out_n, 0, sizeof(uint32_t), st));
        if (d_status) BM_TRY(hipMemsetAsync(d_status, 0, sizeof(uint32_t), st));
        return TRAG_OK;
    }
    return enqueue_query(ix, query_terms, n_terms, k, d_out_docs, d_out_scores, d_out_n, d_status, st);
}

trag_status trag_bm25_search(trag_bm25* ix, const uint32_t* query_terms, uint32_t n_terms, uint32_t k,
                             uint32_t* out_docs, float* out_scores, uint32_t* out_n) {
    if (!ix) return set_error(TRAG_INVALID_ARG, "null index");
    if (!out_n || (n_terms && !query_terms) || (k && (!out_docs || !out_scores)))
        return set_error(TRAG_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    *out_n = 0;
    if (k == 0 || n_terms == 0 || ix->n_live == 0) return TRAG_OK;  // src/index.rs:214-216
    DeviceScope guard(ix->device);
    // any k is legal (the reference truncates): size everything by what can come back
    k = (uint32_t)std::min<uint64_t>(k, ix->n_live);
    trag_status ts = ensure_out(ix, k);
    if (ts != TRAG_OK) return ts;
    hipStream_t st = ix->stream;
    uint32_t* d_docs = ix->d_out;
    float* d_scores = reinterpret_cast<float*>(ix->d_out + k);
    uint32_t* d_n = ix->d_out + 2 * (size_t)k;
    uint32_t* d_status = d_n + 1;
    const uint32_t* tail = ix->h_out + 2 * (size_t)k;  // {count, status} once the block is back
    for (int attempt = 0; attempt < 2; ++attempt) {
        ts = enqueue_query(ix, query_terms, n_terms, k, d_docs, d_scores, d_n, d_status, st);
        if (ts != TRAG_OK) return ts;
        // one transfer into pinned memory brings documents, scores, count and status back
        BM_TRY(hipMemcpyAsync(ix->h_out, ix->d_out, ((size_t)2 * k + 2) * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        BM_TRY(hipStreamSynchronize(st));
        if (tail[1] != ST_NEED_REFINE) break;
        ix->full_select = true;  // crowded coarse bin: use the multi-level select from now on
    }
    if (tail[1] != ST_OK)
        return set_error(TRAG_DEVICE_ERROR, "search could not be completed (status " + std::to_string(tail[1]) + ")");
    const uint32_t n = std::min(tail[0], k);
    if (n) {
        std::memcpy(out_docs, ix->h_out, n * sizeof(uint32_t));
        std::memcpy(out_scores, ix->h_out + k, n * sizeof(float));
    }
    *out_n = n;
    return TRAG_OK;
}

trag_status trag_bm25_fill_synthetic(trag_bm25* ix, uint64_t n_docs, uint32_t vocab, uint32_t tokens_per_doc,
                                     uint64_t seed, uint64_t first_global_doc, uint32_t* out_first_doc) {
    if (!ix) return set_error(TRAG_INVALID_ARG, "null index");
    if (vocab == 0 || vocab >= MAX_TERM_ID) return set_error(TRAG_INVALID_ARG, "vocab out of range");
    std::lock_guard<std::mutex> lk(ix->mu);
    std::vector<double> cdf(vocab);
    double acc = 0.0;
    for (uint32_t r = 0; r < vocab; ++r) {
        acc += 1.0 / (double)(r + 1);
        cdf[r] = acc;
    }
    const double total = acc;
    // generated and added in slabs so that the token buffer stays small
    const uint64_t slab = 1u << 16;
    std::vector<uint32_t> tok;
    std::vector<uint64_t> off;
    uint32_t first_doc = (uint32_t)ix->doc_len.size();
    for (uint64_t d0 = 0; d0 < n_docs; d0 += slab) {
        const uint64_t nd = std::min(slab, n_docs - d0);
        tok.resize(nd * tokens_per_doc);
        off.resize(nd + 1);
        for (uint64_t i = 0; i <= nd; ++i) off[i] = i * tokens_per_doc;
        const unsigned n_thr = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        std::vector<std::thread> pool;
        for (unsigned w = 0; w < n_thr; ++w) {
            pool.emplace_back([&, w]() {
                for (uint64_t i = w; i < nd; i += n_thr) {
                    const uint64_t doc = first_global_doc + d0 + i;
                    for (uint32_t p = 0; p < tokens_per_doc; ++p) {
                        const uint64_t h = mix64(seed ^ mix64(doc * 0x100000001B3ull + p));
                        const double u = (double)(h >> 11) * (1.0 / 9007199254740992.0) * total;
                        const uint32_t r = (uint32_t)(std::upper_bound(cdf.begin(), cdf.end(), u) - cdf.begin());
                        tok[i * tokens_per_doc + p] = std::min(r, vocab - 1);
                    }
                }
            });
        }
        for (auto& th : pool) th.join();
        trag_status ts = add_docs(ix, tok.data(), off.data(), nd, nullptr);
        if (ts != TRAG_OK) return ts;
    }
    if (out_first_doc) *out_first_doc = first_doc;
    return TRAG_OK;
}

trag_status trag_bm25_stats(trag_bm25* ix, uint64_t* postings, uint64_t* terms) {
    if (!ix || !postings || !terms) return set_error(TRAG_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    *postings = ix->live_postings;
    uint64_t t = 0;
    for (uint32_t c : ix->df) t += c != 0;
    *terms = t;
    return TRAG_OK;
}

}  // extern "C"
