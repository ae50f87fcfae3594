// Host side of the C ABI declared in include/trag.h: the device-resident
// vector store (row-major N x d fp32 in HBM, tombstone bitmap, per-handle
// search scratch) and the search pipeline
//     scan -> select_final                       (one workgroup; the usual case)
//     scan -> refine x2 -> compact -> final      (crowded score bins; chosen once a
//                                                 query reported that it needs it)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "trag.h"
#include "trag_internal.h"

using namespace trag;

namespace {

trag_status fail(trag_status st, const std::string& msg) { return trag::set_error(st, msg); }

}  // namespace

namespace {

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            return fail(_e == hipErrorOutOfMemory ? TRAG_OOM : TRAG_DEVICE_ERROR,             \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                   \
        }                                                                                     \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) {
            switched = hipSetDevice(dev) == hipSuccess;
        }
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

}  // namespace

struct trag_store {
    trag_config cfg{};
    int device = 0;
    float* d_rows = nullptr;       // cap_rows x dim
    uint64_t cap_rows = 0;
    uint64_t n_rows = 0;           // rows ever inserted
    uint64_t n_live = 0;
    uint32_t* d_dead = nullptr;    // bitmap, (cap_rows + 31) / 32 words
    std::vector<uint32_t> h_dead;  // host mirror
    float* d_vmax_sq = nullptr;    // 2 floats: largest squared row norm, largest squared bf16 rounding ratio of a row
    float* d_norm2 = nullptr;      // cap_rows squared row norms (batched cosine epilogue)
    uint16_t* d_mirror = nullptr;  // optional bf16 copy of d_rows (cap_rows x dim), kept in step with it
    BatchScratch batch{};
    Bf16Scratch bf16{};
    uint64_t bf16_tiles_cap = 0;   // tile maxima per query the bf16 scratch holds
    uint32_t bf16_q_cap = 0;
    uint64_t batch_rows_cap = 0;   // rows the batch scratch was sized for
    uint32_t batch_q_cap = 0;
    // search scratch (one query in flight per handle)
    Scratch scratch{};
    void* zero_block = nullptr;    // ctl | hist, zero between queries
    size_t zero_bytes = 0;
    uint64_t keys_cap = 0;
    // scratch of a multi-query exact scan: MULTI_QUERIES strided copies of the above
    Scratch mscratch{};
    MultiStride mstride{};
    void* m_zero_block = nullptr;
    uint64_t m_keys_cap = 0;
    const SearchCtl* last_ctl = nullptr;  // control block of the last query enqueued (diagnostics)
    bool bf16_off = false;         // learned: this corpus overflows the bf16 prefilter's candidate lists
    bool mirror_off = false;       // learned: single-query mirror scans keep falling back on this corpus
    uint32_t mirror_tries = 0, mirror_fails = 0;
    bool full_select = false;      // learned: this corpus needs the multi-level select
    uint32_t refine_calls = 0, select_calls = 0;  // calls that needed it / all calls, while it is not the plan
    uint32_t* h_events = nullptr;  // host-pinned word the select kernels bump when a query needs it
    uint64_t one_wg_queries = 0;   // queries enqueued with the one-workgroup select so far
    uint32_t scan_epoch = 0;       // tag of the last scan with sparse key stores (trag_scan_common.h)
    int sparse_keys = -1;          // 1: single-query scans skip the key stores no later stage can read
    unsigned long long* d_sort = nullptr;  // large-k path: rank keys of every row, padded to 2^m
    unsigned long long sort_cap = 0;
    // staging for the host-pointer entry points
    float* d_q = nullptr;
    uint64_t d_q_cap = 0;          // floats
    // results of a host-pointer call live in ONE device block, laid out per call as
    //   rows [nq * k] | scores [nq * k] | counts [nq] | statuses [nq] | control block of the last query,
    // so that one copy into pinned host memory brings everything back
    uint32_t* d_out_block = nullptr;
    uint32_t* h_out_block = nullptr;  // pinned
    uint64_t out_cap_words = 0;
    float* h_q = nullptr;             // pinned staging of the queries
    uint64_t h_q_cap = 0;             // floats
    uint32_t* d_out_rows = nullptr;   // the four views of the current call
    float* d_out_scores = nullptr;
    uint32_t* d_out_n = nullptr;
    uint32_t* d_out_status = nullptr;
    // Small inserts (the chunk-by-chunk `index()` loop of the retrievers, src/retrieve.rs:156-164)
    // go through a pinned staging ring and return without waiting for the GPU; the per-row norms,
    // the norm bound and the bf16 mirror of the rows in [dirty_from, n_rows) are brought up to date
    // by ONE launch before the next search.
    char* h_ring = nullptr;
    size_t ring_bytes = 0, ring_pos = 0;
    uint64_t dirty_from = ~0ull;
    hipStream_t stream = nullptr;
    // The scratch holds one search at a time: when a call arrives on another stream than the one
    // before it (a device-entry call on a caller's stream, then a host-entry call on `stream`, ...)
    // the new stream first waits for everything enqueued on the old one.
    hipStream_t last_stream = nullptr;
    bool has_last = false;
    hipEvent_t order_ev = nullptr;
    // Pipelined single-query searches (trag_store_set_pipelined): two scratch sets alternate, the scan
    // of a query runs on the caller's stream and its one-workgroup select on `side`, beside the next
    // query's scan; sel_done[x] marks the end of the last select that used set x.
    // Host-entry searches from several threads: a thread that finds the handle busy takes one of a few
    // spare search contexts (own stream, scratch set and staging) instead of waiting -- search(&self)
    // is concurrent in the reference (src/index.rs:386). Created on first use.
    struct SearchCtx {
        Scratch sc{};
        void* zero_block = nullptr;
        uint64_t keys_cap = 0;
        hipStream_t stream = nullptr;
        float* d_q = nullptr;
        float* h_q = nullptr;         // pinned
        uint32_t* d_out = nullptr;    // rows [k] | scores [k] | count | status
        uint32_t* h_out = nullptr;    // pinned
        uint64_t out_words = 0;
        bool busy = false;
    };
    static constexpr int POOL = 3;
    SearchCtx pool[POOL];
    std::mutex pool_mu;
    bool pool_closed = false;      // a call that changes the store is running: no search beside it
    bool pipelined = false;
    Scratch scratch2{};
    void* zero_block2 = nullptr;
    uint64_t keys_cap2 = 0;
    hipStream_t side = nullptr;
    hipEvent_t scan_done[2] = {nullptr, nullptr}, sel_done[2] = {nullptr, nullptr};
    bool pipe_used[2] = {false, false};
    uint32_t pipe_next = 0;
    // Deferred selects (trag_store_set_pipelined(s, 2)): the select of a plain single query is not
    // launched behind its scan but kept here, and rides as one extra workgroup in the launch of the
    // NEXT query's scan (other scratch set) -- one launch per query, no event, no second stream. Anything
    // that needs the results first (trag_store_fence, any other route, a change of stream, a scratch
    // set about to be replaced) launches it on its own, on the stream its scan ran on.
    bool deferred = false, has_pend = false;
    PrevSelect pend{};
    hipStream_t pend_stream = nullptr;
    hipEvent_t pend_ev = nullptr;
    std::mutex mu;
    trag_search_stats stats{};
    // optional scan timing
    bool profiling = false;
    uint32_t profile_every = 1;    // bracket every n-th scan launch with events
    uint32_t profile_seen = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;  // created once, reused
    std::vector<uint32_t> ev_launches;   // scan launches between the two events of a pair
    size_t ev_used = 0;
    // deferred selects: a step is ONE launch and the launches follow one another without a gap, which an
    // event pair around a single launch would open (the dispatch can no longer be prepared under the
    // previous kernel: + 2 us on a 237 us scan); the pairs therefore span runs of consecutive launches
    uint32_t span_count = 0, span_len = 0;
    hipEvent_t span_close = nullptr;
    hipStream_t span_stream = nullptr;
};

namespace {

// Held (with s->mu) by every call that changes the store or frees what searches read: no spare
// context is handed out while it lives, and it waits for the ones in use. Searches beside a change
// are the caller's error (&mut self in the reference), but they stay safe, as they were when one
// lock serialised everything.
struct PoolGate {
    trag_store* s;
    bool reopen;
    explicit PoolGate(trag_store* st, bool reopen_ = true) : s(st), reopen(reopen_) {
        for (;;) {
            {
                std::lock_guard<std::mutex> lk(s->pool_mu);
                s->pool_closed = true;
                bool busy = false;
                for (auto& c : s->pool) busy = busy || c.busy;
                if (!busy) return;
            }
            std::this_thread::yield();
        }
    }
    ~PoolGate() {
        if (!reopen) return;
        std::lock_guard<std::mutex> lk(s->pool_mu);
        s->pool_closed = false;
    }
};

trag_status grow_rows(trag_store* s, uint64_t want_rows) {
    if (want_rows <= s->cap_rows) return TRAG_OK;
    if (want_rows > 0xFFFFFFF0ull) return fail(TRAG_INVALID_ARG, "row count exceeds u32 row ids");
    uint64_t cap = std::max<uint64_t>(want_rows, s->cap_rows + s->cap_rows / 2);
    cap = std::max<uint64_t>(cap, 1024);
    cap = (cap + 63) & ~63ull;
    float* nrows = nullptr;
    uint32_t* ndead = nullptr;
    float* nnorm = nullptr;
    const size_t row_bytes = (size_t)cap * s->cfg.dim * sizeof(float);
    const size_t dead_words = (size_t)(cap + 31) / 32;
    HIP_TRY(hipMalloc(&nrows, row_bytes));
    hipError_t e = hipMalloc(&ndead, dead_words * sizeof(uint32_t));
    if (e != hipSuccess) {
        (void)hipFree(nrows);
        return fail(TRAG_OOM, std::string("hipMalloc(dead bitmap): ") + hipGetErrorString(e));
    }
    e = hipMalloc(&nnorm, (size_t)cap * sizeof(float));
    if (e != hipSuccess) {
        (void)hipFree(nrows);
        (void)hipFree(ndead);
        return fail(TRAG_OOM, std::string("hipMalloc(row norms): ") + hipGetErrorString(e));
    }
    uint16_t* nmirror = nullptr;
    if (s->d_mirror) {
        e = hipMalloc(&nmirror, row_bytes / 2);
        if (e != hipSuccess) {
            (void)hipFree(nrows);
            (void)hipFree(ndead);
            (void)hipFree(nnorm);
            return fail(TRAG_OOM, std::string("hipMalloc(bf16 mirror): ") + hipGetErrorString(e));
        }
    }
    e = hipMemsetAsync(ndead, 0, dead_words * sizeof(uint32_t), s->stream);
    if (e == hipSuccess && s->n_rows && nmirror)
        e = hipMemcpyAsync(nmirror, s->d_mirror, (size_t)s->n_rows * s->cfg.dim * sizeof(uint16_t),
                           hipMemcpyDeviceToDevice, s->stream);
    if (e == hipSuccess && s->n_rows)
        e = hipMemcpyAsync(nnorm, s->d_norm2, (size_t)s->n_rows * sizeof(float), hipMemcpyDeviceToDevice, s->stream);
    if (e == hipSuccess && s->n_rows)
        e = hipMemcpyAsync(nrows, s->d_rows, (size_t)s->n_rows * s->cfg.dim * sizeof(float),
                           hipMemcpyDeviceToDevice, s->stream);
    if (e == hipSuccess && s->n_rows)
        e = hipMemcpyAsync(ndead, s->d_dead, ((size_t)(s->n_rows + 31) / 32) * sizeof(uint32_t),
                           hipMemcpyDeviceToDevice, s->stream);
    // (the stream is drained on failure too: the copies above may still read the new buffers)
    const hipError_t e_sync = hipStreamSynchronize(s->stream);
    if (e == hipSuccess) e = e_sync;
    if (e != hipSuccess) {   // the store keeps its old buffers; the new ones go back
        (void)hipFree(nrows);
        (void)hipFree(ndead);
        (void)hipFree(nnorm);
        if (nmirror) (void)hipFree(nmirror);
        return fail(TRAG_DEVICE_ERROR, std::string("growing the store: ") + hipGetErrorString(e));
    }
    if (s->side) (void)hipStreamSynchronize(s->side);
    if (s->d_rows) (void)hipFree(s->d_rows);
    if (s->d_dead) (void)hipFree(s->d_dead);
    if (s->d_norm2) (void)hipFree(s->d_norm2);
    if (s->d_mirror) (void)hipFree(s->d_mirror);
    s->d_mirror = nmirror;
    s->d_rows = nrows;
    s->d_dead = ndead;
    s->d_norm2 = nnorm;
    s->cap_rows = cap;
    s->h_dead.resize(dead_words, 0u);
    return TRAG_OK;
}

// One search scratch set (the handle has one; a pipelined handle two). A set that must grow is only
// replaced once nothing can still be using it: the caller has drained the pipeline.
trag_status ensure_scratch_set(trag_store* s, Scratch& sc, void*& zero_block, uint64_t& keys_cap,
                               hipStream_t init_stream, bool shared_state = true) {
    if (!sc.cand) HIP_TRY(hipMalloc(&sc.cand, CAND_CAP * sizeof(uint32_t)));
    if (!sc.ckeys) HIP_TRY(hipMalloc(&sc.ckeys, CAND_CAP * sizeof(uint32_t)));
    if (shared_state) {   // (a spare context of a concurrent search touches nothing the handle's lock guards)
        if (!s->h_events) {
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_events), 64, hipHostMallocMapped));
            *s->h_events = 0u;
        }
        if (!sc.events) {
            void* dp = nullptr;
            HIP_TRY(hipHostGetDevicePointer(&dp, s->h_events, 0));
            sc.events = static_cast<uint32_t*>(dp);
        }
    }
    if (keys_cap < s->cap_rows || !zero_block) {
        void* old[] = {sc.keys, sc.tmax, zero_block};
        for (void* p : old)
            if (p) (void)hipFree(p);
        sc.keys = nullptr;
        sc.tmax = nullptr;
        zero_block = nullptr;
        keys_cap = 0;
        if (shared_state) s->last_ctl = nullptr;
        const size_t ctl_bytes = 256;  // SearchCtl padded
        static_assert(sizeof(SearchCtl) <= 256, "control block grew");
        const size_t zero_bytes = ctl_bytes + HIST_LEVELS * HIST_BINS * sizeof(uint32_t);
        if (shared_state) s->zero_bytes = zero_bytes;
        HIP_TRY(hipMalloc(&zero_block, zero_bytes));
        // stream-ordered clear, finished before any stream can use the block (a plain
        // hipMemset on the null stream is not ordered against non-blocking streams)
        HIP_TRY(hipMemsetAsync(zero_block, 0, zero_bytes, init_stream));
        HIP_TRY(hipStreamSynchronize(init_stream));
        sc.ctl = reinterpret_cast<SearchCtl*>(zero_block);
        sc.hist = reinterpret_cast<uint32_t*>(static_cast<char*>(zero_block) + ctl_bytes);
        if (!sc.wmax) HIP_TRY(hipMalloc(&sc.wmax, WMAX_CAP * sizeof(uint32_t)));
        if (!sc.wbest) {
            HIP_TRY(hipMalloc(&sc.wbest, SPARSE_WAVES * sizeof(unsigned long long)));
            HIP_TRY(hipMemsetAsync(sc.wbest, 0, SPARSE_WAVES * sizeof(unsigned long long), init_stream));
            HIP_TRY(hipStreamSynchronize(init_stream));
        }
        HIP_TRY(hipMalloc(&sc.keys, (size_t)s->cap_rows * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&sc.tmax, ((size_t)s->cap_rows / 64 + 1) * sizeof(uint32_t)));
        keys_cap = s->cap_rows;
    }
    return TRAG_OK;
}

hipError_t span_timer_close(trag_store* s);

// A deferred select that is still pending: launched now, alone, on the stream its scan ran on.
trag_status flush_pending(trag_store* s) {
    if (!s->has_pend) return TRAG_OK;
    HIP_TRY(span_timer_close(s));   // (a run of timed launches ends where something else enters the stream)
    const PrevSelect& p = s->pend;
    s->has_pend = false;
    Scratch sc{};
    sc.ctl = p.ctl;
    sc.hist = p.hist;
    sc.keys = const_cast<uint32_t*>(p.keys);
    sc.tmax = const_cast<uint32_t*>(p.tmax);
    sc.wmax = const_cast<uint32_t*>(p.wmax);
    sc.events = p.events;
    sc.wbest = const_cast<unsigned long long*>(p.wbest);
    sc.epoch = p.epoch;
    const ScanArgs a{s->d_rows, s->d_dead, p.n_rows, p.d, s->cfg.metric};
    HIP_TRY(launch_select_final(p.q, a, sc, false, p.live, p.need, p.k, p.n_waves, p.out_rows, p.out_scores, p.out_n,
                                p.out_status, s->pend_stream));
    return TRAG_OK;
}

// Makes `st` wait for the selects a pipelined handle still owes: a deferred one is launched (and `st`
// ordered behind it when it is another stream than the scan's), those on the side stream are waited for.
// No-op on a handle that is not pipelined.
trag_status drain_pipeline(trag_store* s, hipStream_t st) {
    if (s->has_pend) {
        const hipStream_t ps = s->pend_stream;
        trag_status ts = flush_pending(s);
        if (ts != TRAG_OK) return ts;
        if (ps != st) {
            HIP_TRY(hipEventRecord(s->pend_ev, ps));
            HIP_TRY(hipStreamWaitEvent(st, s->pend_ev, 0));
        }
    }
    for (int x = 0; x < 2; ++x)
        if (s->pipe_used[x]) HIP_TRY(hipStreamWaitEvent(st, s->sel_done[x], 0));
    return TRAG_OK;
}

trag_status ensure_scratch(trag_store* s) {
    // (a set is replaced when the store has grown: nothing of the pipeline may still be reading it)
    const bool two = s->pipelined || s->deferred;
    const bool replace = s->keys_cap < s->cap_rows || (two && s->keys_cap2 < s->cap_rows);
    if (replace && (s->pipe_used[0] || s->pipe_used[1])) {
        HIP_TRY(hipStreamSynchronize(s->side));
        s->pipe_used[0] = s->pipe_used[1] = false;
    }
    if (replace && s->has_pend) {   // (the pending select reads the set that is about to go)
        const hipStream_t ps = s->pend_stream;
        trag_status tp = flush_pending(s);
        if (tp != TRAG_OK) return tp;
        HIP_TRY(hipStreamSynchronize(ps));
    }
    trag_status ts = ensure_scratch_set(s, s->scratch, s->zero_block, s->keys_cap, s->stream);
    if (ts == TRAG_OK && two) ts = ensure_scratch_set(s, s->scratch2, s->zero_block2, s->keys_cap2, s->stream);
    return ts;
}

trag_status ensure_multi_scratch(trag_store* s) {
    if (s->m_zero_block && s->m_keys_cap >= s->cap_rows) return TRAG_OK;
    void* old[] = {s->mscratch.keys, s->mscratch.tmax, s->mscratch.wmax, s->m_zero_block};
    for (void* p : old)
        if (p) (void)hipFree(p);
    s->mscratch = Scratch{};
    s->m_zero_block = nullptr;
    s->m_keys_cap = 0;
    s->last_ctl = nullptr;
    MultiStride ms;
    ms.zero_bytes = 256 + HIST_LEVELS * HIST_BINS * sizeof(uint32_t);
    ms.keys = (size_t)s->cap_rows;
    ms.tmax = (size_t)s->cap_rows / 64 + 1;
    ms.wmax = WMAX_CAP;
    HIP_TRY(hipMalloc(&s->m_zero_block, ms.zero_bytes * MULTI_QUERIES));
    HIP_TRY(hipMemsetAsync(s->m_zero_block, 0, ms.zero_bytes * MULTI_QUERIES, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipMalloc(&s->mscratch.keys, ms.keys * MULTI_QUERIES * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&s->mscratch.tmax, ms.tmax * MULTI_QUERIES * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&s->mscratch.wmax, ms.wmax * MULTI_QUERIES * sizeof(uint32_t)));
    s->mscratch.events = s->scratch.events;
    s->mscratch.ctl = reinterpret_cast<SearchCtl*>(s->m_zero_block);
    s->mscratch.hist = reinterpret_cast<uint32_t*>(static_cast<char*>(s->m_zero_block) + 256);
    s->mstride = ms;
    s->m_keys_cap = s->cap_rows;
    return TRAG_OK;
}

constexpr uint64_t OUT_CTL_WORDS = 64;  // room for a SearchCtl behind the results (diagnostics)

trag_status ensure_staging(trag_store* s, uint64_t nq, uint64_t k) {
    static_assert(sizeof(SearchCtl) <= OUT_CTL_WORDS * sizeof(uint32_t), "control block grew");
    const uint64_t need_q = nq * s->cfg.dim;
    if (need_q > s->d_q_cap) {
        if (s->d_q) (void)hipFree(s->d_q);
        s->d_q = nullptr;
        s->d_q_cap = 0;
        HIP_TRY(hipMalloc(&s->d_q, need_q * sizeof(float)));
        s->d_q_cap = need_q;
    }
    if (need_q > s->h_q_cap) {
        if (s->h_q) (void)hipHostFree(s->h_q);
        s->h_q = nullptr;
        s->h_q_cap = 0;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_q), need_q * sizeof(float), hipHostMallocDefault));
        s->h_q_cap = need_q;
    }
    const uint64_t kk = std::max<uint64_t>(k, 1);
    const uint64_t words = 2 * nq * kk + 2 * nq + OUT_CTL_WORDS;
    if (words > s->out_cap_words) {
        if (s->d_out_block) (void)hipFree(s->d_out_block);
        if (s->h_out_block) (void)hipHostFree(s->h_out_block);
        s->d_out_block = nullptr;
        s->h_out_block = nullptr;
        s->out_cap_words = 0;
        const uint64_t cap = std::max<uint64_t>(words, 4096);
        HIP_TRY(hipMalloc(&s->d_out_block, cap * sizeof(uint32_t)));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_out_block), cap * sizeof(uint32_t), hipHostMallocDefault));
        s->out_cap_words = cap;
    }
    s->d_out_rows = s->d_out_block;
    s->d_out_scores = reinterpret_cast<float*>(s->d_out_block + nq * kk);
    s->d_out_n = s->d_out_block + 2 * nq * kk;
    s->d_out_status = s->d_out_n + nq;
    return TRAG_OK;
}

trag_status ensure_batch_scratch(trag_store* s, uint32_t nq) {
    const uint32_t qcap = std::min<uint32_t>(std::max<uint32_t>(nq, 1), BATCH_QUERIES);
    if (s->batch.keys && s->batch_rows_cap >= s->cap_rows && s->batch_q_cap >= qcap) return TRAG_OK;
    void* old[] = {s->batch.qpad, s->batch.ctls, s->batch.keys, s->batch.tmax, s->batch.cand};
    for (void* p : old)
        if (p) (void)hipFree(p);
    s->batch = BatchScratch{};
    s->batch_rows_cap = 0;
    s->batch_q_cap = 0;
    const uint32_t qalloc = std::max(qcap, s->batch_q_cap);
    const size_t qpad_rows = ((size_t)qalloc + 127) / 128 * 128;
    s->batch.keys_stride = (size_t)s->cap_rows;
    s->batch.tmax_stride = (size_t)s->cap_rows / 64 + 1;
    HIP_TRY(hipMalloc(&s->batch.qpad, qpad_rows * s->cfg.dim * sizeof(float)));
    HIP_TRY(hipMalloc(&s->batch.ctls, (size_t)qalloc * sizeof(SearchCtl)));
    HIP_TRY(hipMalloc(&s->batch.keys, (size_t)qalloc * s->batch.keys_stride * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&s->batch.tmax, (size_t)qalloc * s->batch.tmax_stride * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&s->batch.cand, (size_t)qalloc * CAND_CAP * sizeof(uint32_t)));
    s->batch_rows_cap = s->cap_rows;
    s->batch_q_cap = qalloc;
    return TRAG_OK;
}

trag_status ensure_bf16_scratch(trag_store* s, uint32_t nq) {
    const uint32_t qcap = std::min<uint32_t>(std::max<uint32_t>(nq, 1), bf16_max_batch(s->cfg.dim, s->cfg.metric, nq));
    const uint64_t tiles = (uint64_t)s->cap_rows / 32 + 32;  // one maximum per 32 rows, sample step 1 at worst
    if (s->bf16.qb && s->bf16_tiles_cap >= tiles && s->bf16_q_cap >= qcap) return TRAG_OK;
    void* old[] = {s->bf16.qb, s->bf16.thr, s->bf16.ctls, s->bf16.tmax, s->bf16.cand, s->bf16.ckeys, s->bf16.rowc};
    for (void* p : old)
        if (p) (void)hipFree(p);
    s->bf16 = Bf16Scratch{};
    s->bf16_tiles_cap = 0;
    s->bf16_q_cap = 0;
    const size_t padded = ((size_t)qcap + 255) / 256 * 256;
    // the sample pass covers at most every tile (step 1) but normally 1/8 of them
    s->bf16.tmax_stride = (size_t)tiles;
    HIP_TRY(hipMalloc(&s->bf16.qb, padded * s->cfg.dim * sizeof(uint16_t)));
    // thresholds, then the ladder state of the tightening thresholds: level, (level 0, step), counters
    HIP_TRY(hipMalloc(&s->bf16.thr, padded * (4 + DYN_LEVELS) * sizeof(float)));
    s->bf16.dyn_lvl = reinterpret_cast<uint32_t*>(s->bf16.thr + padded);
    s->bf16.dyn_lad = s->bf16.thr + 2 * padded;
    s->bf16.dyn_cnt = reinterpret_cast<uint32_t*>(s->bf16.thr + 4 * padded);
    HIP_TRY(hipMalloc(&s->bf16.ctls, (size_t)qcap * sizeof(SearchCtl)));
    HIP_TRY(hipMalloc(&s->bf16.tmax, (size_t)qcap * s->bf16.tmax_stride * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&s->bf16.cand, (size_t)qcap * CAND_CAP * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&s->bf16.ckeys, (size_t)qcap * CAND_CAP * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&s->bf16.rowc, ((size_t)s->cap_rows + 64) * 2 * sizeof(float)));
    s->bf16_tiles_cap = tiles;
    s->bf16_q_cap = qcap;
    return TRAG_OK;
}

// The bf16 prefilter serves batches from 4 queries up on corpora large enough for its
// sample pass to be meaningful; k must leave room in the candidate lists. Its 9/8
// corpus reads cost about as much as a 3-query exact pass and less than a 4-query
// one (10M x 384: 2.50 ms against 2.57 / 2.69 / 3.47 ms for 3 / 4 / 8 queries).
// With the bf16 mirror the pass reads half the bytes (10M x 384: about 1.3 ms) and serves
// batches from 2 queries up. AUTO leaves the prefilter alone where its candidate lists
// would near their capacity (large dimensions: the bound grows with sqrt(d) in units of
// the score spread).
bool use_bf16_path(const trag_store* s, uint32_t mode, uint32_t nq, uint32_t k) {
    if (!bf16_supports(s->cfg.dim, s->cfg.metric) || s->n_rows < 65536 || k > 256) return false;
    if (mode == TRAG_SCAN_BF16) return true;
    if (mode == TRAG_SCAN_MIRROR) return s->d_mirror != nullptr && nq >= 2;
    if (mode != TRAG_SCAN_AUTO || s->bf16_off) return false;
    if (bf16_list_estimate(s->n_rows, k, s->cfg.dim) > 2048.0) return false;
    // Up to 8 queries the alternative is the multi-query exact pass, which costs one scan plus 12 / 21 / 29 / ... / 71 %
    // for 2 / 3 / 4 / ... / 8 queries; the prefilter costs one pass over the rows (or over the mirror: half a scan) plus
    // some 40 us of small launches whatever the batch. Measured at d = 384 (ms per step, multi-query / prefilter): 1 M rows
    // 2 queries 0.266 / 0.275, 3: 0.286 / 0.278, 8: 0.405 / 0.280; 250 k rows 4: 0.089 / 0.107, 8: 0.125 / 0.109; 100 k
    // rows 8: 0.073 / 0.076; 10 M rows 2: 2.49 / 2.22.
    const uint32_t multi_max = multi_group_limit(s->cfg.dim);
    if (nq >= 2 && nq <= 8 && nq <= multi_max) {
        static const double extra[9] = {0.0, 0.0, 0.12, 0.21, 0.29, 0.40, 0.50, 0.61, 0.71};
        const double scan_s = (double)s->n_rows * s->cfg.dim * 4.0 / 6.5e12 + 12e-6;
        return scan_s * (extra[nq] + (s->d_mirror ? 0.5 : 0.0)) > 40e-6;
    }
    return nq >= (s->d_mirror ? 2u : 4u);
}

// Single queries read the bf16 mirror when there is one (half the bytes of the exact scan).
bool use_mirror_scan(const trag_store* s, uint32_t mode, uint32_t nq, uint32_t k) {
    if (!s->d_mirror || nq != 1 || s->mirror_off) return false;
    if (!mirror_scan_supports_dim(s->cfg.dim) || s->n_rows < 65536 || k > 256) return false;
    return mode == TRAG_SCAN_AUTO || mode == TRAG_SCAN_MIRROR;
}

// Batches go to the matrix cores when the mode asks for it, or by default above
// 8 queries, if the metric/dimension have a kernel and the corpus is not tiny.
// Up to 8 queries one multi-query exact pass (HBM-bound) is as fast as the
// matrix-core pass with its smallest (32-query) tile and needs no re-score
// (10M x 384: 8 queries 3.50 ms exact vs 3.63 ms; 16 queries 6.9 ms vs 3.7 ms).
bool use_batch_path(const trag_store* s, uint32_t mode, uint32_t nq) {
    if (!batch_supports(s->cfg.dim, s->cfg.metric) || s->n_rows < 4096) return false;
    if (mode == TRAG_SCAN_MFMA) return true;
    return mode == TRAG_SCAN_AUTO && nq > 8;
}

uint32_t pick_mode(const trag_store* s, uint32_t requested) {
    uint32_t mode = requested;
    // The exact-order scan measured faster than the streaming one on MI355X (it
    // needs no re-score), so it is the default for every dimension.
    if (mode == TRAG_SCAN_AUTO || mode == TRAG_SCAN_MFMA || mode == TRAG_SCAN_BF16 || mode == TRAG_SCAN_MIRROR)
        mode = TRAG_SCAN_EXACT;
    if (mode == TRAG_SCAN_STREAM && !stream_supports_dim(s->cfg.dim)) mode = TRAG_SCAN_EXACT;
    return mode;
}

uint32_t choose_mode(const trag_store* s, uint32_t n_queries, uint32_t k) {
    const uint32_t k_eff = (uint32_t)std::min<uint64_t>(k, s->n_live);
    if (use_mirror_scan(s, s->cfg.scan_mode, n_queries, k_eff)) return TRAG_SCAN_MIRROR;
    if (use_bf16_path(s, s->cfg.scan_mode, n_queries, k_eff)) return TRAG_SCAN_BF16;
    const uint32_t asked = (s->cfg.scan_mode == TRAG_SCAN_BF16 || s->cfg.scan_mode == TRAG_SCAN_MIRROR)
                               ? (uint32_t)TRAG_SCAN_AUTO : s->cfg.scan_mode;
    if (use_batch_path(s, asked, n_queries)) return TRAG_SCAN_MFMA;
    return pick_mode(s, s->cfg.scan_mode);
}

// Optional scan timing: every profile_every-th scan launch is bracketed by an event pair on
// the stream it runs on. Records the opening event and hands back the closing one (nullptr
// when this launch is not timed); the caller records it right after the scan launch.
hipError_t scan_timer_begin(trag_store* s, hipStream_t st, hipEvent_t* closing) {
    *closing = nullptr;
    if (!s->profiling || s->ev_used >= 4096 || (s->profile_seen++ % s->profile_every) != 0) return hipSuccess;
    if (s->ev_used == s->ev_pool.size()) {
        hipEvent_t a0 = nullptr, a1 = nullptr;
        hipError_t e = hipEventCreate(&a0);
        if (e != hipSuccess) return e;
        e = hipEventCreate(&a1);
        if (e != hipSuccess) { (void)hipEventDestroy(a0); return e; }
        s->ev_pool.emplace_back(a0, a1);
    }
    const auto& pair = s->ev_pool[s->ev_used];
    if (s->ev_launches.size() < s->ev_pool.size()) s->ev_launches.resize(s->ev_pool.size(), 1u);
    s->ev_launches[s->ev_used] = 1u;
    s->ev_used += 1;
    *closing = pair.second;
    return hipEventRecord(pair.first, st);
}

// The same for the launches of a handle with deferred selects: one pair around every run of
// max(profile_every, 4) consecutive launches (each launch is timed, as part of its run).
hipError_t span_timer_close(trag_store* s) {
    if (!s->span_close) return hipSuccess;
    const hipError_t e = hipEventRecord(s->span_close, s->span_stream);
    s->ev_launches[s->ev_used - 1] = s->span_count;
    s->span_close = nullptr;
    s->span_count = 0;
    return e;
}
hipError_t span_timer_before_launch(trag_store* s, hipStream_t st) {
    if (!s->profiling) return hipSuccess;
    if (s->span_close && s->span_stream != st) {
        const hipError_t e = span_timer_close(s);
        if (e != hipSuccess) return e;
    }
    if (!s->span_close) {
        const uint32_t every = s->profile_every;
        s->profile_every = 1;
        s->profile_seen = 0;
        hipEvent_t closing = nullptr;
        const hipError_t e = scan_timer_begin(s, st, &closing);
        s->profile_every = every;
        if (e != hipSuccess || !closing) return e;
        s->span_close = closing;
        s->span_stream = st;
        s->span_count = 0;
        s->span_len = std::max<uint32_t>(every, 4u);
    }
    return hipSuccess;
}
hipError_t span_timer_after_launch(trag_store* s) {
    if (!s->span_close) return hipSuccess;
    if (++s->span_count >= s->span_len) return span_timer_close(s);
    return hipSuccess;
}

// Orders work about to be enqueued on `st` behind whatever the handle enqueued last, if that
// went to another stream: an event is recorded at the old stream's tail only then, so a caller
// that stays on one stream pays nothing. Caller holds s->mu. (A stream handed to a _device entry
// must not be destroyed before the handle's next call; if recording on it fails the device is
// drained instead.)
trag_status order_after_last(trag_store* s, hipStream_t st) {
    if (s->has_last && s->last_stream != st) {   // (selects of a pipelined handle run on its side stream)
        trag_status td = drain_pipeline(s, st);
        if (td != TRAG_OK) return td;
    }
    if (s->has_last && s->last_stream != st && s->order_ev) {
        hipError_t e = hipEventRecord(s->order_ev, s->last_stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(st, s->order_ev, 0);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            HIP_TRY(hipDeviceSynchronize());
        }
    }
    s->last_stream = st;
    s->has_last = true;
    return TRAG_OK;
}

// Derived data (squared norms, their bound, the bf16 mirror) of rows that small inserts appended
// since the last search: one launch over the whole range, on the store's stream. Caller holds
// s->mu and calls order_after_last() afterwards, which orders a search on another stream behind it.
trag_status finish_inserts(trag_store* s) {
    if (s->dirty_from >= s->n_rows) { s->dirty_from = ~0ull; return TRAG_OK; }
    const uint64_t r0 = s->dirty_from, n = s->n_rows - r0;
    const uint32_t dim = s->cfg.dim;
    trag_status ts = order_after_last(s, s->stream);
    if (ts != TRAG_OK) return ts;
    float* dst = s->d_rows + (size_t)r0 * dim;
    HIP_TRY(launch_update_vmax(dst, n, dim, s->d_vmax_sq, s->d_norm2 + r0, s->stream));
    if (s->d_mirror) HIP_TRY(launch_to_bf16(dst, n * dim, s->d_mirror + (size_t)r0 * dim, s->stream));
    s->dirty_from = ~0ull;
    return TRAG_OK;
}

// Enqueue the pipeline for n_queries device-resident queries on `st`.
// Caller holds s->mu, has set the device and made sure n_live > 0 and k > 0.

// Tags the next single-query scan for the sparse key stores (the tag tells this scan's
// wbest[] entries from older ones; 0 = off: the multi-level select reads every key).
// TRAG_SPARSE_KEYS=0 turns them off for the process.
static hipError_t next_scan_epoch(trag_store* s, Scratch& sc, bool allowed, hipStream_t st) {
    if (s->sparse_keys < 0) {
        const char* e = getenv("TRAG_SPARSE_KEYS");
        s->sparse_keys = (e && e[0] == '0') ? 0 : 1;
    }
    sc.epoch = 0;
    if (!s->sparse_keys || !allowed || !sc.wbest) return hipSuccess;
    hipError_t err = hipSuccess;
    if (++s->scan_epoch == 0u) {   // wrapped: forget every old tag (of both scratch sets, nothing in flight)
        s->scan_epoch = 1;
        if (s->side && (s->pipe_used[0] || s->pipe_used[1])) err = hipStreamSynchronize(s->side);
        if (err == hipSuccess && s->has_pend) {   // (a deferred select still reads its set's tags)
            const hipStream_t ps = s->pend_stream;
            if (flush_pending(s) != TRAG_OK) err = hipErrorUnknown;
            if (err == hipSuccess) err = hipStreamSynchronize(ps);
        }
        if (err == hipSuccess) err = hipMemsetAsync(s->scratch.wbest, 0, SPARSE_WAVES * sizeof(unsigned long long), st);
        if (err == hipSuccess && s->scratch2.wbest)
            err = hipMemsetAsync(s->scratch2.wbest, 0, SPARSE_WAVES * sizeof(unsigned long long), st);
    }
    sc.epoch = s->scan_epoch;
    return err;
}

// Whether a single-query search on this handle takes the multi-level select, decided per call from what
// the handle has learned (s->full_select) and from the store as it is now -- so the plan follows a store
// that grows, and both device entries (with and without allocations) decide alike.
// Euclidean stores too short for sparse key stores go straight to the multi-level select. Distances
// between normalised embeddings without near neighbours lie within a few per cent of one another:
// nearly every row shares the coarse bin (sign, exponent, two mantissa bits) of the k-th best, the
// one-workgroup select gives up, and a caller of the device entry points, who enqueues far ahead,
// would see that status on every query enqueued before the handle had learned it. The multi-level
// form costs 8 us a query more (10 k rows: 41 against 33 us; 1 M: 0.256 against 0.253 ms); on
// longer scans it would also cost the sparse key stores (10 M x 384: 2.33 against 2.23 ms), so
// those start with the one-workgroup form, which falls through to the multi-level stages inside the
// same enqueue when its bin is crowded (see enqueue_search).
bool plan_full_select(const trag_store* s) {
    if (s->full_select) return true;
    return s->cfg.metric == TRAG_EUCLIDEAN && !sparse_scan_pays((uint32_t)((s->n_rows + 63) / 64), 1024u) &&
           !getenv("TRAG_EUCLID_ONE_WG");
}

trag_status enqueue_search(trag_store* s, const float* d_queries, uint32_t n_queries, uint32_t k,
                           uint32_t mode, uint32_t* d_out_rows, float* d_out_scores,
                           uint32_t* d_out_n, uint32_t* d_status, hipStream_t st, bool may_pipeline = false) {
    trag_status ts = ensure_scratch(s);
    if (ts != TRAG_OK) return ts;
    const uint32_t live = (uint32_t)s->n_live;
    const uint32_t k_eff = std::min<uint32_t>(k, live);
    // Callers of the device entry point never hand a status back, so the handle learns from the
    // word the select kernels bump in pinned host memory (read as it is, no synchronisation):
    // once at least 4 queries, and at least half of those enqueued so far, asked for the
    // multi-level select, it becomes the plan.
    if (!s->full_select && s->h_events) {
        const uint32_t ev = *static_cast<volatile uint32_t*>(s->h_events);
        if (ev >= 4 && (uint64_t)ev * 2 >= s->one_wg_queries) s->full_select = true;
    }
    const bool full_select = plan_full_select(s);
    s->scratch.epoch = 0;  // sparse key stores: only the single-query scans below switch them on
    // Pipelined form (device entry of a handle that asked for it, one exact-order query, the
    // one-workgroup select): every other route runs on `st` alone, behind whatever the pipeline still
    // has on the side stream.
    const bool pipe_base = may_pipeline && ((s->pipelined && s->side) || s->deferred) && n_queries == 1 &&
                           !(k_eff > CAND_CAP - STREAM_SLACK && live > CAND_CAP);
    const bool pipe_any = pipe_base && !full_select;
    const bool pipe = pipe_any && mode == TRAG_SCAN_EXACT;
    // (the mirror scan's three small stages always go to the side stream: they are launches of their own;
    // so do the stages of the multi-level select, the plan of Euclidean stores too short for sparse key stores)
    const bool pipe_mirror = pipe_any && mode == TRAG_SCAN_MIRROR && s->side && s->d_mirror;
    const bool pipe_full = pipe_base && full_select && mode == TRAG_SCAN_EXACT && s->side;
    if (!pipe && !pipe_mirror && !pipe_full && (s->pipe_used[0] || s->pipe_used[1] || s->has_pend)) {
        ts = drain_pipeline(s, st);
        if (ts != TRAG_OK) return ts;
    }
    if (k_eff > CAND_CAP - STREAM_SLACK && live > CAND_CAP) {
        // More results than the one-workgroup sort holds: exact scan, then a
        // global sort of every row's rank key (rare; the reference's callers ask
        // for at most 100).
        unsigned long long padded = 64;
        while (padded < s->n_rows) padded <<= 1;
        if (s->sort_cap < padded) {
            if (s->d_sort) (void)hipFree(s->d_sort);
            s->d_sort = nullptr;
            s->sort_cap = 0;
            HIP_TRY(hipMalloc(&s->d_sort, padded * sizeof(unsigned long long)));
            s->sort_cap = padded;
        }
        ScanArgs a{s->d_rows, s->d_dead, (uint32_t)s->n_rows, s->cfg.dim, s->cfg.metric};
        for (uint32_t b = 0; b < n_queries; ++b) {
            const float* q = d_queries + (size_t)b * s->cfg.dim;
            uint32_t n_waves = 0;
            HIP_TRY(launch_scan_exact(q, a, s->scratch, k_eff, k_eff, s->d_vmax_sq, &n_waves, st));
            HIP_TRY(launch_full_sort(a, s->scratch, s->d_sort, padded, k_eff, d_out_rows + (size_t)b * k,
                                     d_out_scores + (size_t)b * k, d_out_n + b,
                                     d_status ? d_status + b : nullptr, st));
            HIP_TRY(hipMemsetAsync(s->scratch.hist, 0, s->zero_bytes - 256, st));
        }
        s->stats.mode_used = TRAG_SCAN_EXACT;
        return TRAG_OK;
    }
    if (mode == TRAG_SCAN_BF16) {
        ts = ensure_bf16_scratch(s, n_queries);
        if (ts != TRAG_OK) return ts;
        ScanArgs a{s->d_rows, s->d_dead, (uint32_t)s->n_rows, s->cfg.dim, s->cfg.metric};
        // one corpus pass per chunk: the query-stationary form takes the whole batch (up to 2048
        // queries) in one, the tiled form 256 at a time
        const uint32_t chunk = bf16_max_batch(s->cfg.dim, s->cfg.metric, n_queries);
        for (uint32_t b0 = 0; b0 < n_queries; b0 += chunk) {
            const uint32_t nb = std::min<uint32_t>(chunk, n_queries - b0);
            const float* Qb = d_queries + (size_t)b0 * s->cfg.dim;
            hipEvent_t e1 = nullptr;
            HIP_TRY(scan_timer_begin(s, st, &e1));
            HIP_TRY(launch_bf16_scan(Qb, nb, a, s->d_mirror, s->d_norm2, s->bf16, k_eff, k_eff, s->d_vmax_sq, st));
            if (e1) HIP_TRY(hipEventRecord(e1, st));
            HIP_TRY(launch_bf16_finish(Qb, nb, a, s->bf16, live, k, d_out_rows + (size_t)b0 * k,
                                       d_out_scores + (size_t)b0 * k, d_out_n + b0,
                                       d_status ? d_status + b0 : nullptr, st));
        }
        s->stats.mode_used = mode;
        return TRAG_OK;
    }
    if (mode == TRAG_SCAN_MFMA) {
        const uint32_t need_b = (uint32_t)std::min<uint64_t>((uint64_t)k_eff + STREAM_SLACK, live);
        if (need_b > CAND_CAP) return fail(TRAG_UNSUPPORTED, "k exceeds the single-workgroup final sort");
        ts = ensure_batch_scratch(s, n_queries);
        if (ts != TRAG_OK) return ts;
        ScanArgs a{s->d_rows, s->d_dead, (uint32_t)s->n_rows, s->cfg.dim, s->cfg.metric};
        for (uint32_t b0 = 0; b0 < n_queries; b0 += BATCH_QUERIES) {
            const uint32_t nb = std::min<uint32_t>(BATCH_QUERIES, n_queries - b0);
            const float* Qb = d_queries + (size_t)b0 * s->cfg.dim;
            hipEvent_t e1 = nullptr;
            HIP_TRY(scan_timer_begin(s, st, &e1));
            HIP_TRY(launch_batch_scan(Qb, nb, a, s->d_norm2, s->batch, need_b, k_eff, s->d_vmax_sq, st));
            if (e1) HIP_TRY(hipEventRecord(e1, st));
            HIP_TRY(launch_batch_finish(Qb, nb, a, s->batch, live, k, d_out_rows + (size_t)b0 * k,
                                        d_out_scores + (size_t)b0 * k, d_out_n + b0,
                                        d_status ? d_status + b0 : nullptr, st));
        }
        s->stats.mode_used = mode;
        return TRAG_OK;
    }
    if (mode == TRAG_SCAN_MIRROR) {
        // one query at a time: scan of the bf16 mirror, then the one-workgroup prune + exact scoring
        if (!s->d_mirror) return fail(TRAG_INVALID_CONFIG, "the bf16 mirror is not enabled on this store");
        // one-workgroup form: some slack below the k-th key so that the coarse cut clears it;
        // multi-level form: the k-th approximate key itself is resolved, then lowered by the bound
        const uint32_t need_m = full_select ? k_eff
                                               : (uint32_t)std::min<uint64_t>((uint64_t)k_eff + STREAM_SLACK, live);
        ScanArgs a{s->d_rows, s->d_dead, (uint32_t)s->n_rows, s->cfg.dim, s->cfg.metric};
        if (pipe_mirror) {
            // scan of the mirror on the caller's stream; prune, exact scores and ordering on the side stream
            // beside the NEXT query's scan (two scratch sets; the query must stay valid until the fence)
            const uint32_t x = s->pipe_next;
            s->pipe_next ^= 1u;
            Scratch& sc = x ? s->scratch2 : s->scratch;
            if (s->has_pend) {   // (a deferred select of the exact route: launched now, ahead of this scan)
                ts = s->pend_stream == st ? flush_pending(s) : drain_pipeline(s, st);
                if (ts != TRAG_OK) return ts;
            }
            if (s->pipe_used[x]) HIP_TRY(hipStreamWaitEvent(st, s->sel_done[x], 0));
            hipEvent_t e1 = nullptr;
            HIP_TRY(scan_timer_begin(s, st, &e1));
            uint32_t n_waves = 0;
            s->last_ctl = sc.ctl;
            HIP_TRY(next_scan_epoch(s, sc, true, st));
            HIP_TRY(launch_scan_mirror(s->d_mirror, d_queries, s->d_norm2, a, sc, need_m, k_eff, s->d_vmax_sq, &n_waves, st));
            if (e1) HIP_TRY(hipEventRecord(e1, st));
            HIP_TRY(hipEventRecord(s->scan_done[x], st));
            HIP_TRY(hipStreamWaitEvent(s->side, s->scan_done[x], 0));
            s->one_wg_queries += 1;
            HIP_TRY(launch_select_mirror(d_queries, a, sc, false, live, need_m, k_eff, n_waves, d_out_rows, d_out_scores,
                                         d_out_n, d_status, s->side));
            HIP_TRY(hipEventRecord(s->sel_done[x], s->side));
            s->pipe_used[x] = true;
            s->stats.mode_used = mode;
            return TRAG_OK;
        }
        for (uint32_t b = 0; b < n_queries; ++b) {
            const float* q = d_queries + (size_t)b * s->cfg.dim;
            hipEvent_t e1 = nullptr;
            HIP_TRY(scan_timer_begin(s, st, &e1));
            uint32_t n_waves = 0;
            HIP_TRY(next_scan_epoch(s, s->scratch, !full_select, st));
            HIP_TRY(launch_scan_mirror(s->d_mirror, q, s->d_norm2, a, s->scratch, need_m, k_eff, s->d_vmax_sq,
                                       &n_waves, st));
            if (e1) HIP_TRY(hipEventRecord(e1, st));
            if (!full_select) s->one_wg_queries += 1;
            HIP_TRY(launch_select_mirror(q, a, s->scratch, full_select, live, need_m, k_eff, n_waves,
                                         d_out_rows + (size_t)b * k, d_out_scores + (size_t)b * k, d_out_n + b,
                                         d_status ? d_status + b : nullptr, st));
            // (the one-workgroup prune clears the coarse histogram itself)
            if (full_select) HIP_TRY(hipMemsetAsync(s->scratch.hist, 0, s->zero_bytes - 256, st));
        }
        s->last_ctl = s->scratch.ctl;
        s->stats.mode_used = mode;
        return TRAG_OK;
    }
    const bool rescore = mode == TRAG_SCAN_STREAM;
    uint64_t need64 = rescore ? (uint64_t)k_eff + STREAM_SLACK : k_eff;
    const uint32_t need = (uint32_t)std::min<uint64_t>(need64, live);
    if (need > CAND_CAP)
        return fail(TRAG_UNSUPPORTED, "k exceeds the single-workgroup final sort (k <= " +
                                          std::to_string(CAND_CAP - STREAM_SLACK) +
                                          " unless the store holds fewer rows)");
    ScanArgs a{s->d_rows, s->d_dead, (uint32_t)s->n_rows, s->cfg.dim, s->cfg.metric};
    // Small batches on the exact path: up to MULTI_QUERIES queries share one corpus pass.
    const uint32_t multi_max = multi_group_limit(s->cfg.dim);
    const bool multi_ok = !rescore && !full_select && multi_max >= 2 && n_queries >= 2;
    if (multi_ok) {
        ts = ensure_multi_scratch(s);
        if (ts != TRAG_OK) return ts;
    }
    for (uint32_t b = 0; b < n_queries; ++b) {
        const float* q = d_queries + (size_t)b * s->cfg.dim;
        hipEvent_t e1 = nullptr;
        if (!(pipe && s->deferred)) HIP_TRY(scan_timer_begin(s, st, &e1));   // (deferred launches are timed in runs)
        uint32_t n_waves = 0;
        const uint32_t group = multi_ok ? std::min<uint32_t>(multi_max, n_queries - b) : 1u;
        if (group >= 2) {
            HIP_TRY(launch_scan_exact_multi(q, group, a, s->mscratch, s->mstride, need, k_eff, s->d_vmax_sq,
                                            &n_waves, st));
            if (e1) HIP_TRY(hipEventRecord(e1, st));
            s->one_wg_queries += group;
            HIP_TRY(launch_select_final(q, a, s->mscratch, false, live, need, k_eff, n_waves,
                                        d_out_rows + (size_t)b * k, d_out_scores + (size_t)b * k, d_out_n + b,
                                        d_status ? d_status + b : nullptr, st, group, &s->mstride, k));
            s->last_ctl = reinterpret_cast<const SearchCtl*>(reinterpret_cast<const char*>(s->mscratch.ctl) +
                                                             (size_t)(group - 1) * s->mstride.zero_bytes);
            b += group - 1;
            continue;
        }
        uint32_t* o_rows = d_out_rows + (size_t)b * k;
        float* o_scores = d_out_scores + (size_t)b * k;
        uint32_t* o_status = d_status ? d_status + b : nullptr;
        if (pipe && s->deferred) {
            // this query's scan, with the previous query's select as one more workgroup of the launch;
            // this query's select is kept for the next launch (or for whoever needs the results first)
            const uint32_t x = s->pipe_next;
            s->pipe_next ^= 1u;
            Scratch& sc = x ? s->scratch2 : s->scratch;
            if (s->has_pend && (s->pend.ctl == sc.ctl || s->pend_stream != st || !scan_can_fuse(s->cfg.dim))) {
                // (cannot ride along: same set, another stream -- order_after_last has ordered the streams
                // by now --, or a scan form without the extra workgroup)
                ts = drain_pipeline(s, st);
                if (ts != TRAG_OK) return ts;
            }
            if (s->pipe_used[x]) {   // (a mirror-scan query's stages on the side stream used this set last)
                HIP_TRY(hipStreamWaitEvent(st, s->sel_done[x], 0));
                s->pipe_used[x] = false;
            }
            s->last_ctl = sc.ctl;
            HIP_TRY(next_scan_epoch(s, sc, true, st));
            HIP_TRY(span_timer_before_launch(s, st));
            HIP_TRY(launch_scan_exact(q, a, sc, need, k_eff, s->d_vmax_sq, &n_waves, st, s->has_pend ? &s->pend : nullptr));
            HIP_TRY(span_timer_after_launch(s));
            s->one_wg_queries += 1;
            PrevSelect& p = s->pend;
            p.q = q;
            p.n_rows = a.n_rows;
            p.d = a.dim;
            p.live = live;
            p.need = need;
            p.k = k_eff;
            p.n_waves = n_waves;
            p.ctl = sc.ctl;
            p.hist = sc.hist;
            p.keys = sc.keys;
            p.tmax = sc.tmax;
            p.wmax = sc.wmax;
            p.out_rows = o_rows;
            p.out_scores = o_scores;
            p.out_n = d_out_n + b;
            p.out_status = o_status;
            p.events = sc.events;
            p.wbest = sc.wbest;
            p.epoch = sc.wbest ? sc.epoch : 0u;
            s->has_pend = true;
            s->pend_stream = st;
            continue;
        }
        if (pipe) {
            // scan on the caller's stream, select on the side stream beside the NEXT query's scan; the
            // set's previous select (two queries ago) must have finished before this scan writes into it
            const uint32_t x = s->pipe_next;
            s->pipe_next ^= 1u;
            Scratch& sc = x ? s->scratch2 : s->scratch;
            if (s->pipe_used[x]) HIP_TRY(hipStreamWaitEvent(st, s->sel_done[x], 0));
            s->last_ctl = sc.ctl;
            HIP_TRY(next_scan_epoch(s, sc, true, st));
            HIP_TRY(launch_scan_exact(q, a, sc, need, k_eff, s->d_vmax_sq, &n_waves, st));
            if (e1) HIP_TRY(hipEventRecord(e1, st));
            HIP_TRY(hipEventRecord(s->scan_done[x], st));
            HIP_TRY(hipStreamWaitEvent(s->side, s->scan_done[x], 0));
            s->one_wg_queries += 1;
            HIP_TRY(launch_select_final(q, a, sc, false, live, need, k_eff, n_waves, o_rows, o_scores, d_out_n + b,
                                        o_status, s->side));
            HIP_TRY(hipEventRecord(s->sel_done[x], s->side));
            s->pipe_used[x] = true;
            continue;
        }
        if (pipe_full) {
            // exact scan on the caller's stream; radix levels, compaction, ordering and the clearing of the
            // accumulators on the side stream beside the next query's scan
            const uint32_t x = s->pipe_next;
            s->pipe_next ^= 1u;
            Scratch& sc = x ? s->scratch2 : s->scratch;
            if (s->has_pend) {
                ts = s->pend_stream == st ? flush_pending(s) : drain_pipeline(s, st);
                if (ts != TRAG_OK) return ts;
            }
            if (s->pipe_used[x]) HIP_TRY(hipStreamWaitEvent(st, s->sel_done[x], 0));
            s->last_ctl = sc.ctl;
            HIP_TRY(next_scan_epoch(s, sc, false, st));
            HIP_TRY(launch_scan_exact(q, a, sc, need, k_eff, s->d_vmax_sq, &n_waves, st));
            if (e1) HIP_TRY(hipEventRecord(e1, st));
            HIP_TRY(hipEventRecord(s->scan_done[x], st));
            HIP_TRY(hipStreamWaitEvent(s->side, s->scan_done[x], 0));
            HIP_TRY(launch_select(a, sc, s->side));
            HIP_TRY(launch_final(q, a, sc, false, live, o_rows, o_scores, d_out_n + b, o_status, s->side));
            HIP_TRY(hipMemsetAsync(sc.hist, 0, s->zero_bytes - 256, s->side));
            HIP_TRY(launch_reset_ctl(sc, s->side));
            HIP_TRY(hipEventRecord(s->sel_done[x], s->side));
            s->pipe_used[x] = true;
            continue;
        }
        s->last_ctl = s->scratch.ctl;
        HIP_TRY(next_scan_epoch(s, s->scratch, !rescore && !full_select, st));
        if (rescore) HIP_TRY(launch_scan_stream(q, a, s->scratch, need, k_eff, s->d_vmax_sq, &n_waves, st));
        else HIP_TRY(launch_scan_exact(q, a, s->scratch, need, k_eff, s->d_vmax_sq, &n_waves, st));
        if (e1) HIP_TRY(hipEventRecord(e1, st));
        if (!full_select) {
            s->one_wg_queries += 1;
            HIP_TRY(launch_select_final(q, a, s->scratch, rescore, live, need, k_eff, n_waves, o_rows,
                                        o_scores, d_out_n + b, o_status, st));
        } else {
            HIP_TRY(launch_select(a, s->scratch, st));
            HIP_TRY(launch_final(q, a, s->scratch, rescore, live, o_rows, o_scores, d_out_n + b, o_status, st));
            // only the accumulators need clearing: ctl keeps the diagnostics until the next scan
            HIP_TRY(hipMemsetAsync(s->scratch.hist, 0, s->zero_bytes - 256, st));
            HIP_TRY(launch_reset_ctl(s->scratch, st));
        }
    }
    s->stats.mode_used = mode;
    return TRAG_OK;
}

}  // namespace

// ---------------------------------------------------------------------------
// trag_store_search_device_noalloc: the search scratch in a fixed device array
// ---------------------------------------------------------------------------
// Callers that cannot allocate (HIP graph capture, serving loops with a frozen memory plan) get the
// single-query pipeline with its scratch -- control block + histograms, wave maxima, candidate
// lists, published wave bests, tile maxima and the per-row keys -- laid out in ONE device array of
// fixed size that every GPU running this library holds from the load of the code object to the end
// of the process: 4 bytes per row for up to NOALLOC_MAX_ROWS rows, about 65.3 MB in all. Every
// call on a GPU shares that GPU's copy, whatever the store: calls that overlap on one GPU race.
namespace {
constexpr uint64_t NOALLOC_MAX_ROWS = 16ull << 20;   // 16 Mi rows: the 10M-row headline corpus fits
constexpr size_t NA_CTL_WORDS = 64;                  // the control block padded to 256 bytes
constexpr size_t NA_ZERO_WORDS = NA_CTL_WORDS + (size_t)HIST_LEVELS * HIST_BINS;
constexpr size_t NA_WMAX = NA_ZERO_WORDS;
constexpr size_t NA_CAND = NA_WMAX + WMAX_CAP;
constexpr size_t NA_CKEYS = NA_CAND + CAND_CAP;
constexpr size_t NA_WBEST = NA_CKEYS + CAND_CAP;     // 64-bit entries: an even word offset
constexpr size_t NA_TMAX = NA_WBEST + 2 * (size_t)SPARSE_WAVES;
constexpr size_t NA_KEYS = NA_TMAX + NOALLOC_MAX_ROWS / 64 + 64;
constexpr size_t NA_WORDS = NA_KEYS + NOALLOC_MAX_ROWS;
static_assert(sizeof(SearchCtl) <= NA_CTL_WORDS * 4, "control block grew");
static_assert(NA_WBEST % 2 == 0 && NA_TMAX % 64 == 0 && NA_KEYS % 64 == 0, "sections lose their alignment");
}  // namespace

static __device__ __align__(256) uint32_t scratch[NA_WORDS];

namespace {

// One entry per device ordinal: the device address of that GPU's copy of `scratch` (each GPU has
// its own, at its own address), looked up once with that device current, and the tag counter of
// the sparse key stores (the published wave bests in the array are shared by every handle).
struct NoAllocDevice {
    uint32_t* base = nullptr;
    hipError_t lookup = hipErrorNotInitialized;
};
constexpr int NOALLOC_MAX_DEVICES = 64;
NoAllocDevice g_noalloc[NOALLOC_MAX_DEVICES];
std::mutex g_noalloc_mu;

// Device `dev` must be current. Host code never uses the array's name as an address: only the
// pointer hipGetSymbolAddress returns for this device.
trag_status noalloc_setup(int dev) {
    if (dev < 0 || dev >= NOALLOC_MAX_DEVICES) return fail(TRAG_UNSUPPORTED, "device ordinal beyond the no-allocation table");
    std::lock_guard<std::mutex> lk(g_noalloc_mu);
    NoAllocDevice& nd = g_noalloc[dev];
    if (nd.base) return TRAG_OK;
    void* p = nullptr;
    nd.lookup = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (nd.lookup != hipSuccess || !p) {
        (void)hipGetLastError();
        return fail(TRAG_DEVICE_ERROR, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(nd.lookup));
    }
    nd.base = static_cast<uint32_t*>(p);
    return TRAG_OK;
}

}  // namespace

extern "C" {

trag_status trag_store_search_device_noalloc(trag_store* s, const float* d_queries, uint32_t n_queries,
                                             uint32_t dim, uint32_t k, uint32_t* d_out_rows,
                                             float* d_out_scores, uint32_t* d_out_n, uint32_t* d_status,
                                             void* hip_stream) {
    if (!s) return fail(TRAG_INVALID_ARG, "null store");
    if (dim != s->cfg.dim)
        return fail(TRAG_DIM_MISMATCH, "dimension mismatch: expected " + std::to_string(s->cfg.dim) +
                                           ", got " + std::to_string(dim));
    if (n_queries == 0) return TRAG_OK;
    if (!d_queries || !d_out_n || (k && (!d_out_rows || !d_out_scores)))
        return fail(TRAG_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    DeviceGuard guard(s->device);
    hipStream_t st = (hipStream_t)hip_stream;
    if (k == 0 || s->n_live == 0) {   // (a kernel, as every write of this entry: see launch_zero_words)
        HIP_TRY(launch_zero_words(d_out_n, n_queries, d_status, d_status ? n_queries : 0u, st));
        return TRAG_OK;
    }
    if (s->n_rows > NOALLOC_MAX_ROWS)
        return fail(TRAG_UNSUPPORTED, "store holds " + std::to_string(s->n_rows) + " rows, the fixed scratch of this "
                                      "entry point " + std::to_string(NOALLOC_MAX_ROWS) + " at most");
    const uint32_t live = (uint32_t)s->n_live;
    const uint32_t k_eff = std::min<uint32_t>(k, live);
    if (k_eff > CAND_CAP - STREAM_SLACK && live > CAND_CAP)
        return fail(TRAG_UNSUPPORTED, "k above " + std::to_string(CAND_CAP - STREAM_SLACK) +
                                      " needs the global sort's buffer (use trag_store_search_device)");
    trag_status ts = noalloc_setup(s->device);   // (a lookup only, and only if create could not do it)
    if (ts != TRAG_OK) return ts;
    std::lock_guard<std::mutex> lk2(g_noalloc_mu);   // one enqueue at a time per process
    NoAllocDevice& nd = g_noalloc[s->device];
    Scratch sc{};
    sc.ctl = reinterpret_cast<SearchCtl*>(nd.base);
    sc.hist = nd.base + NA_CTL_WORDS;
    sc.wmax = nd.base + NA_WMAX;
    sc.cand = nd.base + NA_CAND;
    sc.ckeys = nd.base + NA_CKEYS;
    sc.wbest = reinterpret_cast<unsigned long long*>(nd.base + NA_WBEST);
    sc.tmax = nd.base + NA_TMAX;
    sc.keys = nd.base + NA_KEYS;
    sc.events = nullptr;   // no pinned word here: this entry follows the plan the handle has, it learns none
    // Under stream capture the call is recorded, not run: work on other streams cannot be ordered
    // against it here (the handle's own stream is not part of the capture), and nothing the handle
    // remembers about its last call describes the replays. Rows appended by small inserts must have
    // been brought up to date by an earlier search; the caller orders replays against the handle's
    // other calls.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (st && hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    if (cap == hipStreamCaptureStatusActive) {
        if (s->dirty_from < s->n_rows)
            return fail(TRAG_UNSUPPORTED, "rows were inserted since the last search: run one search outside the capture first");
    } else {
        ts = finish_inserts(s);   // (kernel launches only)
        if (ts == TRAG_OK) ts = order_after_last(s, st);
        if (ts != TRAG_OK) return ts;
    }
    // The array keeps what the last call left, possibly of another store, of a call that failed half
    // way, or -- a captured call replayed -- of this very call: the accumulators and the published wave
    // bests start from zero in every call (keys, tile and wave maxima are rewritten by every scan for
    // every row they are read for). The tag that tells the bests of one scan from another's is the
    // query's number inside the call: a tag that survived the call would be baked into a captured graph
    // and every replay would take the previous replay's bests for its own.
    if (s->sparse_keys < 0) {
        const char* e = getenv("TRAG_SPARSE_KEYS");
        s->sparse_keys = (e && e[0] == '0') ? 0 : 1;
    }
    const bool full_select = plan_full_select(s);
    const bool sparse = s->sparse_keys && !full_select;
    HIP_TRY(launch_zero_words(nd.base, (uint32_t)NA_ZERO_WORDS, reinterpret_cast<uint32_t*>(sc.wbest),
                              sparse ? 2u * SPARSE_WAVES : 0u, st));
    const size_t hist_bytes = (size_t)HIST_LEVELS * HIST_BINS * sizeof(uint32_t);
    ScanArgs a{s->d_rows, s->d_dead, (uint32_t)s->n_rows, s->cfg.dim, s->cfg.metric};
    for (uint32_t b = 0; b < n_queries; ++b) {
        const float* q = d_queries + (size_t)b * dim;
        sc.epoch = sparse ? b + 1u : 0u;
        uint32_t n_waves = 0;
        HIP_TRY(launch_scan_exact(q, a, sc, k_eff, k_eff, s->d_vmax_sq, &n_waves, st));
        uint32_t* o_rows = d_out_rows + (size_t)b * k;
        float* o_scores = d_out_scores + (size_t)b * k;
        uint32_t* o_status = d_status ? d_status + b : nullptr;
        if (!full_select) {
            HIP_TRY(launch_select_final(q, a, sc, false, live, k_eff, k_eff, n_waves, o_rows, o_scores,
                                        d_out_n + b, o_status, st));
        } else {
            HIP_TRY(launch_select(a, sc, st));
            HIP_TRY(launch_final(q, a, sc, false, live, o_rows, o_scores, d_out_n + b, o_status, st));
            HIP_TRY(launch_zero_words(sc.hist, (uint32_t)(hist_bytes / sizeof(uint32_t)), nullptr, 0u, st));
            HIP_TRY(launch_reset_ctl(sc, st));
        }
    }
    s->stats.mode_used = TRAG_SCAN_EXACT;
    return TRAG_OK;
}

const char* trag_version(void) { return "trag-hip 0.1 (gfx950)"; }

trag_status trag_store_create(const trag_config* cfg, trag_store** out) {
    if (!cfg || !out) return fail(TRAG_INVALID_ARG, "null argument");
    *out = nullptr;
    if (cfg->dim == 0) return fail(TRAG_INVALID_CONFIG, "dimension must be positive");
    if (cfg->metric > TRAG_DOT_PRODUCT) return fail(TRAG_INVALID_CONFIG, "unknown metric");
    if (cfg->scan_mode > TRAG_SCAN_MIRROR) return fail(TRAG_INVALID_CONFIG, "unknown scan mode");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(TRAG_DEVICE_ERROR, "no HIP device available (this library has no CPU path)");
    int dev = cfg->device;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    if (dev >= ndev) return fail(TRAG_INVALID_CONFIG, "device ordinal out of range");
    DeviceGuard guard(dev);
    trag_store* s = new trag_store();
    s->cfg = *cfg;
    s->cfg.device = dev;
    s->device = dev;
    e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete s; return fail(TRAG_DEVICE_ERROR, hipGetErrorString(e)); }
    e = hipEventCreateWithFlags(&s->order_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc(&s->d_vmax_sq, 2 * sizeof(float));
    if (e == hipSuccess) e = hipMemsetAsync(s->d_vmax_sq, 0, 2 * sizeof(float), s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    if (e != hipSuccess) { trag_store_destroy(s); return fail(TRAG_DEVICE_ERROR, hipGetErrorString(e)); }
    if (cfg->reserve_rows) {
        trag_status ts = grow_rows(s, cfg->reserve_rows);
        if (ts != TRAG_OK) { trag_store_destroy(s); return ts; }
    }
    // set-up of the fixed scratch of trag_store_search_device_noalloc on this device (its address is
    // looked up once per device; a failure is reported by that entry point, not here)
    (void)noalloc_setup(dev);
    *out = s;
    return TRAG_OK;
}

void trag_store_destroy(trag_store* s) {
    if (!s) return;
    PoolGate gate(s, false);
    DeviceGuard guard(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    (void)hipDeviceSynchronize();
    if (s->side) (void)hipStreamSynchronize(s->side);
    for (auto& ev : s->ev_pool) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    if (s->order_ev) (void)hipEventDestroy(s->order_ev);
    if (s->pend_ev) (void)hipEventDestroy(s->pend_ev);
    for (int x = 0; x < 2; ++x) {
        if (s->scan_done[x]) (void)hipEventDestroy(s->scan_done[x]);
        if (s->sel_done[x]) (void)hipEventDestroy(s->sel_done[x]);
    }
    for (auto& c : s->pool) {
        if (c.stream) (void)hipStreamSynchronize(c.stream);
        void* cp[] = {c.zero_block, c.sc.cand, c.sc.ckeys, c.sc.keys, c.sc.tmax, c.sc.wmax, c.sc.wbest, c.d_q, c.d_out};
        for (void* p : cp)
            if (p) (void)hipFree(p);
        if (c.h_q) (void)hipHostFree(c.h_q);
        if (c.h_out) (void)hipHostFree(c.h_out);
        if (c.stream) (void)hipStreamDestroy(c.stream);
    }
    void* set2[] = {s->zero_block2, s->scratch2.cand, s->scratch2.ckeys, s->scratch2.keys, s->scratch2.tmax,
                    s->scratch2.wmax, s->scratch2.wbest};
    for (void* p : set2)
        if (p) (void)hipFree(p);
    if (s->side) (void)hipStreamDestroy(s->side);
    void* ptrs[] = {s->d_rows, s->d_dead, s->d_vmax_sq, s->d_norm2, s->d_mirror,
                    s->batch.qpad, s->batch.ctls, s->batch.keys, s->batch.tmax, s->batch.cand,
                    s->bf16.qb, s->bf16.thr, s->bf16.ctls, s->bf16.tmax, s->bf16.cand, s->bf16.ckeys, s->bf16.rowc,
                    s->zero_block, s->scratch.cand, s->scratch.ckeys, s->scratch.keys, s->scratch.tmax, s->scratch.wmax, s->scratch.wbest,
                    s->m_zero_block, s->mscratch.keys, s->mscratch.tmax, s->mscratch.wmax,
                    s->d_sort, s->d_q, s->d_out_block};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (s->h_events) (void)hipHostFree(s->h_events);
    if (s->h_ring) (void)hipHostFree(s->h_ring);
    if (s->h_q) (void)hipHostFree(s->h_q);
    if (s->h_out_block) (void)hipHostFree(s->h_out_block);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

trag_status trag_store_config(const trag_store* s, trag_config* out) {
    if (!s || !out) return fail(TRAG_INVALID_ARG, "null argument");
    *out = s->cfg;
    return TRAG_OK;
}

uint64_t trag_store_len(const trag_store* s) { return s ? s->n_live : 0; }
uint64_t trag_store_rows(const trag_store* s) { return s ? s->n_rows : 0; }

trag_status trag_store_reserve(trag_store* s, uint64_t rows) {
    if (!s) return fail(TRAG_INVALID_ARG, "null store");
    std::lock_guard<std::mutex> lk(s->mu);
    PoolGate gate(s);
    DeviceGuard guard(s->device);
    return grow_rows(s, rows);
}

trag_status trag_store_insert(trag_store* s, const float* rows, uint64_t n, uint32_t dim,
                              uint32_t* out_first_row) {
    if (!s) return fail(TRAG_INVALID_ARG, "null store");
    if (n > 0 && !rows) return fail(TRAG_INVALID_CONFIG, "chunk must have embedding");
    if (dim != s->cfg.dim)
        return fail(TRAG_DIM_MISMATCH, "dimension mismatch: expected " + std::to_string(s->cfg.dim) +
                                           ", got " + std::to_string(dim));
    std::lock_guard<std::mutex> lk(s->mu);
    PoolGate gate(s);
    DeviceGuard guard(s->device);
    if (out_first_row) *out_first_row = (uint32_t)s->n_rows;
    if (n == 0) return TRAG_OK;
    trag_status ts = grow_rows(s, s->n_rows + n);
    if (ts != TRAG_OK) return ts;
    float* dst = s->d_rows + (size_t)s->n_rows * dim;
    const size_t bytes = (size_t)n * dim * sizeof(float);
    if (!s->h_ring) {
        const size_t want = std::max<size_t>(1u << 20, (size_t)64 * dim * sizeof(float));
        if (hipHostMalloc(reinterpret_cast<void**>(&s->h_ring), want, hipHostMallocDefault) == hipSuccess) s->ring_bytes = want;
        else { s->h_ring = nullptr; (void)hipGetLastError(); }
    }
    if (s->h_ring && bytes <= s->ring_bytes / 4) {
        // small insert: through the pinned ring, no wait (the ring's oldest bytes are reused only
        // after the copies that read them have finished: one synchronisation per lap)
        if (s->ring_pos + bytes > s->ring_bytes) {
            HIP_TRY(hipStreamSynchronize(s->stream));
            s->ring_pos = 0;
        }
        trag_status to = order_after_last(s, s->stream);   // (a search may still be reading the corpus tail's neighbours: cheap, keeps one order)
        if (to != TRAG_OK) return to;
        std::memcpy(s->h_ring + s->ring_pos, rows, bytes);
        HIP_TRY(hipMemcpyAsync(dst, s->h_ring + s->ring_pos, bytes, hipMemcpyHostToDevice, s->stream));
        s->ring_pos += (bytes + 255) & ~(size_t)255;
        s->dirty_from = std::min<uint64_t>(s->dirty_from, s->n_rows);
        s->n_rows += n;
        s->n_live += n;
        return TRAG_OK;
    }
    HIP_TRY(hipMemcpyAsync(dst, rows, (size_t)n * dim * sizeof(float), hipMemcpyHostToDevice, s->stream));
    HIP_TRY(launch_update_vmax(dst, n, dim, s->d_vmax_sq, s->d_norm2 + s->n_rows, s->stream));
    if (s->d_mirror)
        HIP_TRY(launch_to_bf16(dst, n * dim, s->d_mirror + (size_t)s->n_rows * dim, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->n_rows += n;
    s->n_live += n;
    return TRAG_OK;
}

trag_status trag_store_overwrite(trag_store* s, uint32_t row, const float* vec, uint32_t dim) {
    if (!s) return fail(TRAG_INVALID_ARG, "null store");
    if (!vec) return fail(TRAG_INVALID_CONFIG, "chunk must have embedding");
    if (dim != s->cfg.dim)
        return fail(TRAG_DIM_MISMATCH, "dimension mismatch: expected " + std::to_string(s->cfg.dim) +
                                           ", got " + std::to_string(dim));
    std::lock_guard<std::mutex> lk(s->mu);
    PoolGate gate(s);
    if (row >= s->n_rows) return fail(TRAG_INVALID_ARG, "row out of range");
    if ((s->h_dead[row >> 5] >> (row & 31)) & 1u) return fail(TRAG_INVALID_ARG, "row was removed");
    DeviceGuard guard(s->device);
    float* dst = s->d_rows + (size_t)row * dim;
    HIP_TRY(hipMemcpyAsync(dst, vec, (size_t)dim * sizeof(float), hipMemcpyHostToDevice, s->stream));
    HIP_TRY(launch_update_vmax(dst, 1, dim, s->d_vmax_sq, s->d_norm2 + row, s->stream));
    if (s->d_mirror) HIP_TRY(launch_to_bf16(dst, dim, s->d_mirror + (size_t)row * dim, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return TRAG_OK;
}

trag_status trag_store_remove(trag_store* s, uint32_t row) {
    if (!s) return fail(TRAG_INVALID_ARG, "null store");
    std::lock_guard<std::mutex> lk(s->mu);
    PoolGate gate(s);
    if (row >= s->n_rows) return fail(TRAG_INVALID_ARG, "row out of range");
    uint32_t& w = s->h_dead[row >> 5];
    if ((w >> (row & 31)) & 1u) return fail(TRAG_INVALID_ARG, "row already removed");
    w |= 1u << (row & 31);
    DeviceGuard guard(s->device);
    HIP_TRY(hipMemcpyAsync(s->d_dead + (row >> 5), &w, sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->n_live -= 1;
    return TRAG_OK;
}

trag_status trag_store_fill_synthetic(trag_store* s, uint64_t n, uint64_t seed,
                                      uint64_t first_global_row, uint32_t* out_first_row) {
    return trag::store_fill_synthetic(s, n, seed, first_global_row, 1, out_first_row);
}

}  // extern "C"

trag_status trag::store_fill_synthetic(trag_store* s, uint64_t n, uint64_t seed, uint64_t first_global_row,
                                       uint64_t row_step, uint32_t* out_first_row) {
    if (!s) return fail(TRAG_INVALID_ARG, "null store");
    std::lock_guard<std::mutex> lk(s->mu);
    PoolGate gate(s);
    DeviceGuard guard(s->device);
    if (out_first_row) *out_first_row = (uint32_t)s->n_rows;
    if (n == 0) return TRAG_OK;
    trag_status ts = grow_rows(s, s->n_rows + n);
    if (ts != TRAG_OK) return ts;
    float* d_norm = nullptr;
    HIP_TRY(hipMalloc(&d_norm, (size_t)n * sizeof(float)));
    float* dst = s->d_rows + (size_t)s->n_rows * s->cfg.dim;
    hipError_t e = launch_fill_synthetic(dst, first_global_row, n, s->cfg.dim, seed, d_norm, s->stream, row_step);
    if (e == hipSuccess) e = launch_update_vmax(dst, n, s->cfg.dim, s->d_vmax_sq, s->d_norm2 + s->n_rows, s->stream);
    if (e == hipSuccess && s->d_mirror)
        e = launch_to_bf16(dst, n * s->cfg.dim, s->d_mirror + (size_t)s->n_rows * s->cfg.dim, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    (void)hipFree(d_norm);
    if (e != hipSuccess) return fail(TRAG_DEVICE_ERROR, std::string("fill_synthetic: ") + hipGetErrorString(e));
    s->n_rows += n;
    s->n_live += n;
    return TRAG_OK;
}

namespace {

// Host-entry search of ONE query by a thread that found the handle busy: runs in a spare context
// (own stream, scratch set, staging) without the handle's lock, beside whatever the lock's holder has
// on the GPU. Takes only the plain route -- exact-order scan, one-workgroup select, nothing pending,
// nothing to learn -- and answers false for everything else (and when no context is free or a query
// does not finish there): the caller then waits for the lock and takes the usual path. Reads of the
// store's rows and counts need no lock: insert / overwrite / remove need exclusive access by contract
// (&mut self in the reference), so no search runs beside them.
bool search_one_concurrent(trag_store* s, const float* query, uint32_t dim, uint32_t k, uint32_t* out_rows,
                           float* out_scores, uint32_t* out_n, trag_status* result) {
    const uint64_t live64 = s->n_live;
    if (k == 0 || live64 == 0 || s->dirty_from < s->n_rows || s->profiling) return false;
    const uint32_t live = (uint32_t)live64;
    const uint32_t k_eff = std::min<uint32_t>(k, live);
    if (k_eff > CAND_CAP - STREAM_SLACK) return false;
    if (choose_mode(s, 1, k) != TRAG_SCAN_EXACT || plan_full_select(s)) return false;
    trag_store::SearchCtx* cx = nullptr;
    {
        std::lock_guard<std::mutex> lk(s->pool_mu);
        if (!s->pool_closed)
            for (auto& c : s->pool)
                if (!c.busy) { c.busy = true; cx = &c; break; }
    }
    if (!cx) return false;
    auto release = [&]() { std::lock_guard<std::mutex> lk(s->pool_mu); cx->busy = false; };
    DeviceGuard guard(s->device);
    auto run = [&]() -> trag_status {
        if (!cx->stream) HIP_TRY(hipStreamCreateWithFlags(&cx->stream, hipStreamNonBlocking));
        trag_status ts = ensure_scratch_set(s, cx->sc, cx->zero_block, cx->keys_cap, cx->stream, false);
        if (ts != TRAG_OK) return ts;
        if (!cx->d_q) {
            HIP_TRY(hipMalloc(&cx->d_q, (size_t)dim * sizeof(float)));
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&cx->h_q), (size_t)dim * sizeof(float), hipHostMallocDefault));
        }
        const uint64_t words = 2ull * k_eff + 2;
        if (words > cx->out_words) {
            if (cx->d_out) (void)hipFree(cx->d_out);
            if (cx->h_out) (void)hipHostFree(cx->h_out);
            cx->d_out = cx->h_out = nullptr;
            cx->out_words = 0;
            const uint64_t cap = std::max<uint64_t>(words, 256);
            HIP_TRY(hipMalloc(&cx->d_out, cap * sizeof(uint32_t)));
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&cx->h_out), cap * sizeof(uint32_t), hipHostMallocDefault));
            cx->out_words = cap;
        }
        std::memcpy(cx->h_q, query, (size_t)dim * sizeof(float));
        HIP_TRY(hipMemcpyAsync(cx->d_q, cx->h_q, (size_t)dim * sizeof(float), hipMemcpyHostToDevice, cx->stream));
        ScanArgs a{s->d_rows, s->d_dead, (uint32_t)s->n_rows, s->cfg.dim, s->cfg.metric};
        cx->sc.epoch = 0;   // every key stored: no tag to manage beside the handle's own scans
        uint32_t n_waves = 0;
        uint32_t* d_rows = cx->d_out;
        float* d_scores = reinterpret_cast<float*>(cx->d_out + k_eff);
        uint32_t* d_n = cx->d_out + 2 * (size_t)k_eff;
        HIP_TRY(launch_scan_exact(cx->d_q, a, cx->sc, k_eff, k_eff, s->d_vmax_sq, &n_waves, cx->stream));
        HIP_TRY(launch_select_final(cx->d_q, a, cx->sc, false, live, k_eff, k_eff, n_waves, d_rows, d_scores, d_n,
                                    d_n + 1, cx->stream));
        HIP_TRY(hipMemcpyAsync(cx->h_out, cx->d_out, words * sizeof(uint32_t), hipMemcpyDeviceToHost, cx->stream));
        HIP_TRY(hipStreamSynchronize(cx->stream));
        return TRAG_OK;
    };
    const trag_status ts = run();
    if (ts != TRAG_OK) { release(); *result = ts; return true; }
    const uint32_t* h = cx->h_out;
    const uint32_t n = h[2 * (size_t)k_eff], status = h[2 * (size_t)k_eff + 1];
    if (status != ST_OK) { release(); return false; }   // (crowded bin: the locked path repeats it and learns)
    std::memcpy(out_rows, h, (size_t)n * sizeof(uint32_t));
    std::memcpy(out_scores, h + k_eff, (size_t)n * sizeof(float));
    *out_n = n;
    release();
    *result = TRAG_OK;
    return true;
}

}  // namespace

bool trag::store_select_pending(trag_store* s) {
    std::lock_guard<std::mutex> lk(s->mu);
    return s->has_pend;
}

trag_status trag::store_wait_side(trag_store* s, hipStream_t st) {
    std::lock_guard<std::mutex> lk(s->mu);
    DeviceGuard guard(s->device);
    for (int x = 0; x < 2; ++x)
        if (s->pipe_used[x]) HIP_TRY(hipStreamWaitEvent(st, s->sel_done[x], 0));
    return TRAG_OK;
}

extern "C" {

trag_status trag_store_read_rows(const trag_store* s, uint64_t row0, uint64_t n, float* out) {
    if (!s || (!out && n)) return fail(TRAG_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(const_cast<trag_store*>(s)->mu);
    if (row0 + n > s->n_rows || row0 + n < row0) return fail(TRAG_INVALID_ARG, "row range out of bounds");
    if (n == 0) return TRAG_OK;
    DeviceGuard guard(s->device);
    HIP_TRY(hipStreamSynchronize(s->stream));   // small inserts are asynchronous copies on it
    HIP_TRY(hipMemcpy(out, s->d_rows + (size_t)row0 * s->cfg.dim, (size_t)n * s->cfg.dim * sizeof(float),
                      hipMemcpyDeviceToHost));
    return TRAG_OK;
}

trag_status trag_store_search_device(trag_store* s, const float* d_queries, uint32_t n_queries,
                                     uint32_t dim, uint32_t k, uint32_t* d_out_rows,
                                     float* d_out_scores, uint32_t* d_out_n, uint32_t* d_status,
                                     void* hip_stream) {
    if (!s) return fail(TRAG_INVALID_ARG, "null store");
    if (dim != s->cfg.dim)
        return fail(TRAG_DIM_MISMATCH, "dimension mismatch: expected " + std::to_string(s->cfg.dim) +
                                           ", got " + std::to_string(dim));
    if (n_queries == 0) return TRAG_OK;
    if (!d_queries || !d_out_n || (k && (!d_out_rows || !d_out_scores)))
        return fail(TRAG_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    DeviceGuard guard(s->device);
    hipStream_t st = (hipStream_t)hip_stream;
    if (k == 0 || s->n_live == 0) {
        HIP_TRY(hipMemsetAsync(d_out_n, 0, n_queries * sizeof(uint32_t), st));
        if (d_status) HIP_TRY(hipMemsetAsync(d_status, 0, n_queries * sizeof(uint32_t), st));
        return TRAG_OK;
    }
    trag_status ts = finish_inserts(s);
    if (ts == TRAG_OK) ts = order_after_last(s, st);
    if (ts != TRAG_OK) return ts;
    const uint32_t mode = choose_mode(s, n_queries, k);
    return enqueue_search(s, d_queries, n_queries, k, mode, d_out_rows, d_out_scores, d_out_n,
                          d_status, st, true);
}

trag_status trag_store_set_pipelined(trag_store* s, int enabled) {
    if (!s) return fail(TRAG_INVALID_ARG, "null store");
    std::lock_guard<std::mutex> lk(s->mu);
    DeviceGuard guard(s->device);
    if (enabled != 0 && enabled != 1 && enabled != 2) return fail(TRAG_INVALID_ARG, "pipelining mode must be 0, 1 or 2");
    if (enabled != 0 && !s->side) {
        HIP_TRY(hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking));
        for (int x = 0; x < 2; ++x) {
            HIP_TRY(hipEventCreateWithFlags(&s->scan_done[x], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&s->sel_done[x], hipEventDisableTiming));
        }
    }
    if (enabled == 2 && !s->pend_ev) HIP_TRY(hipEventCreateWithFlags(&s->pend_ev, hipEventDisableTiming));
    if (enabled == 0 && s->side && (s->pipe_used[0] || s->pipe_used[1])) {
        HIP_TRY(hipStreamSynchronize(s->side));   // (what was enqueued finishes; later calls run on one stream)
        s->pipe_used[0] = s->pipe_used[1] = false;
    }
    if (enabled != 2 && s->has_pend) {   // (a select still owed is launched now, behind its scan)
        trag_status ts = flush_pending(s);
        if (ts != TRAG_OK) return ts;
    }
    s->pipelined = enabled == 1;
    s->deferred = enabled == 2;
    return TRAG_OK;
}

trag_status trag_store_fence(trag_store* s, void* hip_stream) {
    if (!s) return fail(TRAG_INVALID_ARG, "null store");
    std::lock_guard<std::mutex> lk(s->mu);
    DeviceGuard guard(s->device);
    return drain_pipeline(s, (hipStream_t)hip_stream);
}

trag_status trag_store_search_batch(trag_store* s, const float* queries, uint32_t n_queries,
                                    uint32_t dim, uint32_t k, uint32_t* out_rows, float* out_scores,
                                    uint32_t* out_n) {
    if (!s) return fail(TRAG_INVALID_ARG, "null store");
    if (dim != s->cfg.dim)
        return fail(TRAG_DIM_MISMATCH, "dimension mismatch: expected " + std::to_string(s->cfg.dim) +
                                           ", got " + std::to_string(dim));
    if (n_queries == 0) return TRAG_OK;
    if (!queries || !out_n || (k && (!out_rows || !out_scores)))
        return fail(TRAG_INVALID_ARG, "null argument");
    std::unique_lock<std::mutex> lk(s->mu, std::try_to_lock);
    if (!lk.owns_lock()) {   // another thread is searching: one query may run beside it (see above)
        trag_status tc = TRAG_OK;
        if (n_queries == 1 && search_one_concurrent(s, queries, dim, k, out_rows, out_scores, out_n, &tc)) return tc;
        lk.lock();
    }
    if (k == 0 || s->n_live == 0) {
        for (uint32_t b = 0; b < n_queries; ++b) out_n[b] = 0;
        return TRAG_OK;
    }
    DeviceGuard guard(s->device);
    // the reference truncates to k, so any k is legal ("everything" = a huge k): buffers, kernels and
    // the copy back are sized by the rows that can come back; the caller's arrays keep their stride k
    const uint32_t k_user = k;
    k = (uint32_t)std::min<uint64_t>(k, s->n_live);
    trag_status ts = ensure_staging(s, n_queries, k);
    if (ts != TRAG_OK) return ts;
    hipStream_t st = s->stream;
    ts = finish_inserts(s);
    if (ts == TRAG_OK) ts = order_after_last(s, st);
    if (ts != TRAG_OK) return ts;
    // queries go up through pinned memory (a truly asynchronous copy), everything comes back in
    // ONE copy of the result block into pinned memory: rows, scores, counts, statuses and the
    // control block the diagnostics read (five pageable copies cost ~60 us per call)
    std::memcpy(s->h_q, queries, (size_t)n_queries * dim * sizeof(float));
    HIP_TRY(hipMemcpyAsync(s->d_q, s->h_q, (size_t)n_queries * dim * sizeof(float), hipMemcpyHostToDevice, st));
    uint32_t mode = choose_mode(s, n_queries, k);
    ts = enqueue_search(s, s->d_q, n_queries, k, mode, s->d_out_rows, s->d_out_scores, s->d_out_n,
                        s->d_out_status, st);
    if (ts != TRAG_OK) return ts;
    std::vector<uint32_t> status(n_queries);
    SearchCtl ctl{};
    const size_t nk_words = (size_t)n_queries * k, res_words = 2 * nk_words + 2 * (size_t)n_queries;
    auto fetch = [&](uint32_t m) -> trag_status {
        const void* ctl_src = m == TRAG_SCAN_BF16 ? (const void*)s->bf16.ctls
                              : m == TRAG_SCAN_MFMA ? (const void*)s->batch.ctls
                              : (const void*)(s->last_ctl ? s->last_ctl : s->scratch.ctl);
        HIP_TRY(hipMemcpyAsync(s->d_out_block + res_words, ctl_src, sizeof(SearchCtl), hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(s->h_out_block, s->d_out_block, (res_words + OUT_CTL_WORDS) * sizeof(uint32_t),
                               hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        const uint32_t* h = s->h_out_block;
        if (k_user == k) {
            std::memcpy(out_rows, h, nk_words * sizeof(uint32_t));
            std::memcpy(out_scores, h + nk_words, nk_words * sizeof(float));
        } else {
            for (uint32_t b = 0; b < n_queries; ++b) {
                std::memcpy(out_rows + (size_t)b * k_user, h + (size_t)b * k, k * sizeof(uint32_t));
                std::memcpy(out_scores + (size_t)b * k_user, h + nk_words + (size_t)b * k, k * sizeof(float));
            }
        }
        std::memcpy(out_n, h + 2 * nk_words, n_queries * sizeof(uint32_t));
        std::memcpy(status.data(), h + 2 * nk_words + n_queries, n_queries * sizeof(uint32_t));
        std::memcpy(&ctl, h + res_words, sizeof(SearchCtl));
        return TRAG_OK;
    };
    ts = fetch(mode);
    if (ts != TRAG_OK) return ts;
    s->stats.fallbacks = 0;
    // A corpus whose scores cluster (many rows within the bf16 bound of the k-th best)
    // overflows the prefilter's lists for many queries at once: repeat the whole batch on
    // the f32 matrix cores, whose bound is ~50 times tighter, and remember it for AUTO.
    if (mode == TRAG_SCAN_BF16 && n_queries >= 8) {
        uint32_t bad = 0;
        for (uint32_t b = 0; b < n_queries; ++b) bad += status[b] != ST_OK;
        if (bad > n_queries / 8 && use_batch_path(s, TRAG_SCAN_MFMA, n_queries)) {
            if (s->cfg.scan_mode == TRAG_SCAN_AUTO) s->bf16_off = true;
            mode = TRAG_SCAN_MFMA;
            s->stats.fallbacks += bad;
            ts = enqueue_search(s, s->d_q, n_queries, k, mode, s->d_out_rows, s->d_out_scores, s->d_out_n,
                                s->d_out_status, st);
            if (ts != TRAG_OK) return ts;
            ts = fetch(mode);
            if (ts != TRAG_OK) return ts;
        }
    }
    // Repeat what the fast paths could not finish. A crowded coarse bin
    // (ST_NEED_REFINE) needs the multi-level select, which is then remembered
    // for later queries on this handle; an uncertified or overflowing
    // streaming / matrix-core result is repeated with the exact scan.
    const bool was_full = s->full_select;
    bool any_refine = false, retried = false;
    for (uint32_t b = 0; b < n_queries; ++b) {
        uint32_t cur_mode = mode;
        for (int attempt = 0; attempt < 3 && status[b] != ST_OK; ++attempt) {
            if (status[b] == ST_NEED_REFINE) {
                any_refine = true;
                s->full_select = true;
                // (a mirror scan is repeated as such: it has a multi-level form)
                if (cur_mode == TRAG_SCAN_MFMA || cur_mode == TRAG_SCAN_BF16) cur_mode = TRAG_SCAN_EXACT;
            } else {
                s->stats.fallbacks += 1;
                cur_mode = TRAG_SCAN_EXACT;
            }
            ts = enqueue_search(s, s->d_q + (size_t)b * dim, 1, k, cur_mode,
                                s->d_out_rows + (size_t)b * k, s->d_out_scores + (size_t)b * k,
                                s->d_out_n + b, s->d_out_status + b, st);
            if (ts != TRAG_OK) return ts;
            ts = fetch(cur_mode);  // (rare path: the whole block again, this query's slots now final)
            if (ts != TRAG_OK) return ts;
            retried = true;
        }
        if (status[b] != ST_OK)
            return fail(TRAG_DEVICE_ERROR, "search could not be completed (status " +
                                               std::to_string(status[b]) + ")");
    }
    // One odd query (a zero vector: every cosine ties) must not slow the handle down for
    // good: the multi-level select becomes the plan once at least 4 calls, and at least
    // half of all calls so far, needed it.
    if (!was_full) {
        s->select_calls += 1;
        if (any_refine) s->refine_calls += 1;
        s->full_select = s->refine_calls >= 4 && s->refine_calls * 2 >= s->select_calls;
    }
    if (mode == TRAG_SCAN_MIRROR) {
        // a corpus whose scores crowd within the mirror's bound keeps sending queries back to
        // the exact scan (each costs a mirror pass extra): AUTO gives the mirror scan up then
        s->mirror_tries += n_queries;
        s->mirror_fails += s->stats.fallbacks;
        if (s->cfg.scan_mode == TRAG_SCAN_AUTO && s->mirror_tries >= 16 && s->mirror_fails * 2 > s->mirror_tries)
            s->mirror_off = true;
    }
    s->stats.mode_used = mode;
    if (retried && (mode == TRAG_SCAN_MFMA || mode == TRAG_SCAN_BF16)) {
        ts = fetch(mode);  // the diagnostics describe the batch's own pass, not a repeated query's
        if (ts != TRAG_OK) return ts;
    }
    if (mode == TRAG_SCAN_MFMA || mode == TRAG_SCAN_BF16) {   // (ctl: the copy that came back with the results)
        s->stats.candidates = ctl.cand_count;
        s->stats.refine_levels = 0;
    } else {
        s->stats.candidates = ctl.expect;
        s->stats.refine_levels = ctl.levels;
    }
    s->stats.keys_kept_from = (mode == TRAG_SCAN_EXACT || mode == TRAG_SCAN_MIRROR) ? ctl.keep_key : 0u;
    return TRAG_OK;
}

trag_status trag_store_search(trag_store* s, const float* query, uint32_t dim, uint32_t k,
                              uint32_t* out_rows, float* out_scores, uint32_t* out_n) {
    return trag_store_search_batch(s, query, 1, dim, k, out_rows, out_scores, out_n);
}

trag_status trag_store_set_scan_mode(trag_store* s, uint32_t scan_mode) {
    if (!s) return fail(TRAG_INVALID_ARG, "null store");
    if (scan_mode > TRAG_SCAN_MIRROR) return fail(TRAG_INVALID_CONFIG, "unknown scan mode");
    std::lock_guard<std::mutex> lk(s->mu);
    s->cfg.scan_mode = scan_mode;
    return TRAG_OK;
}

trag_status trag_store_set_bf16_mirror(trag_store* s, int enabled) {
    if (!s) return fail(TRAG_INVALID_ARG, "null store");
    std::lock_guard<std::mutex> lk(s->mu);
    PoolGate gate(s);
    DeviceGuard guard(s->device);
    if (!enabled) {
        if (s->d_mirror) {
            HIP_TRY(hipStreamSynchronize(s->stream));
            HIP_TRY(hipDeviceSynchronize());
            (void)hipFree(s->d_mirror);
            s->d_mirror = nullptr;
        }
        return TRAG_OK;
    }
    if (s->d_mirror) return TRAG_OK;
    if (s->cfg.dim % 8u) return fail(TRAG_UNSUPPORTED, "the bf16 mirror needs a dimension that is a multiple of 8");
    {
        trag_status tf = finish_inserts(s);
        if (tf != TRAG_OK) return tf;
    }
    const uint64_t cap = std::max<uint64_t>(s->cap_rows, 64);
    HIP_TRY(hipMalloc(&s->d_mirror, (size_t)cap * s->cfg.dim * sizeof(uint16_t)));
    if (s->cap_rows == 0) {
        // nothing stored yet: grow_rows allocates the real mirror with the first rows
        trag_status ts = grow_rows(s, 64);
        if (ts != TRAG_OK) return ts;
    }
    if (s->n_rows) HIP_TRY(launch_to_bf16(s->d_rows, s->n_rows * s->cfg.dim, s->d_mirror, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->mirror_off = false;
    s->mirror_tries = s->mirror_fails = 0;
    return TRAG_OK;
}

int trag_store_has_bf16_mirror(const trag_store* s) { return s && s->d_mirror ? 1 : 0; }

trag_status trag_store_clone(trag_store* src, trag_store** out) {
    if (!src || !out) return fail(TRAG_INVALID_ARG, "null argument");
    *out = nullptr;
    trag_config cfg;
    uint64_t n_rows = 0;
    {
        std::lock_guard<std::mutex> lk(src->mu);
        cfg = src->cfg;
        n_rows = src->n_rows;
    }
    cfg.reserve_rows = n_rows;
    trag_store* dst = nullptr;
    trag_status ts = trag_store_create(&cfg, &dst);
    if (ts != TRAG_OK) return ts;
    std::lock_guard<std::mutex> lk(src->mu);
    DeviceGuard guard(src->device);
    auto give_up = [&](trag_status code) { trag_store_destroy(dst); return code; };
    if (src->n_rows > dst->cap_rows) {   // rows arrived between the two locks
        ts = grow_rows(dst, src->n_rows);
        if (ts != TRAG_OK) return give_up(ts);
    }
    ts = finish_inserts(src);            // norms of rows that small inserts appended
    if (ts != TRAG_OK) return give_up(ts);
    const uint64_t n = src->n_rows;
    hipError_t e = hipStreamSynchronize(src->stream);
    if (e == hipSuccess && src->has_last && src->last_stream != src->stream) e = hipDeviceSynchronize();
    if (e == hipSuccess && n) {
        const size_t dim = src->cfg.dim;
        e = hipMemcpyAsync(dst->d_rows, src->d_rows, n * dim * sizeof(float), hipMemcpyDeviceToDevice, dst->stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(dst->d_norm2, src->d_norm2, n * sizeof(float), hipMemcpyDeviceToDevice, dst->stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(dst->d_dead, src->d_dead, ((n + 31) / 32) * sizeof(uint32_t), hipMemcpyDeviceToDevice,
                               dst->stream);
    }
    if (e == hipSuccess)
        e = hipMemcpyAsync(dst->d_vmax_sq, src->d_vmax_sq, 2 * sizeof(float), hipMemcpyDeviceToDevice, dst->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(dst->stream);
    if (e != hipSuccess) return give_up(fail(TRAG_DEVICE_ERROR, std::string("clone: ") + hipGetErrorString(e)));
    std::copy(src->h_dead.begin(), src->h_dead.begin() + std::min(src->h_dead.size(), dst->h_dead.size()),
              dst->h_dead.begin());
    dst->n_rows = n;
    dst->n_live = src->n_live;
    const bool mirror = src->d_mirror != nullptr;
    if (mirror) {                        // (takes dst's own lock; nobody else knows dst yet)
        ts = trag_store_set_bf16_mirror(dst, 1);
        if (ts != TRAG_OK) return give_up(ts);
    }
    *out = dst;
    return TRAG_OK;
}

trag_status trag_store_stream_probe(trag_store* s, uint32_t repeats, double* gb_per_s) {
    if (!s || !gb_per_s) return fail(TRAG_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    *gb_per_s = 0.0;
    if (s->n_rows == 0) return TRAG_OK;
    DeviceGuard guard(s->device);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(hipEventCreate(&e0));
    hipError_t e = hipEventCreate(&e1);
    if (e != hipSuccess) { (void)hipEventDestroy(e0); return fail(TRAG_DEVICE_ERROR, hipGetErrorString(e)); }
    const uint64_t n_floats = s->n_rows * s->cfg.dim;
    float best_ms = 0.0f;
    for (uint32_t r = 0; r < std::max<uint32_t>(repeats, 1) && e == hipSuccess; ++r) {
        e = hipEventRecord(e0, s->stream);
        if (e == hipSuccess) e = launch_stream_probe(s->d_rows, n_floats, s->d_vmax_sq, s->stream);  // (the sink is never written)
        if (e == hipSuccess) e = hipEventRecord(e1, s->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.0f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e == hipSuccess && (best_ms == 0.0f || ms < best_ms)) best_ms = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (e != hipSuccess) return fail(TRAG_DEVICE_ERROR, std::string("stream probe: ") + hipGetErrorString(e));
    if (best_ms > 0.0f) *gb_per_s = (double)(n_floats / 4 * 16) / (best_ms * 1e-3) / 1e9;
    return TRAG_OK;
}

trag_status trag_store_set_profiling(trag_store* s, int enabled) {
    if (!s) return fail(TRAG_INVALID_ARG, "null store");
    std::lock_guard<std::mutex> lk(s->mu);
    {
        DeviceGuard guard(s->device);
        HIP_TRY(span_timer_close(s));
    }
    s->profiling = enabled != 0;
    s->profile_every = enabled > 1 ? (uint32_t)enabled : 1u;
    s->profile_seen = 0;
    s->ev_used = 0;
    return TRAG_OK;
}

trag_status trag_store_scan_time(trag_store* s, double* total_ms, uint32_t* launches) {
    if (!s || !total_ms || !launches) return fail(TRAG_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    DeviceGuard guard(s->device);
    HIP_TRY(span_timer_close(s));
    double sum = 0.0;
    uint32_t n = 0;
    for (size_t i = 0; i < s->ev_used; ++i) {
        HIP_TRY(hipEventSynchronize(s->ev_pool[i].second));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, s->ev_pool[i].first, s->ev_pool[i].second));
        sum += ms;
        n += i < s->ev_launches.size() ? s->ev_launches[i] : 1u;
    }
    *total_ms = sum;
    *launches = n;
    s->ev_used = 0;
    return TRAG_OK;
}

trag_status trag_merge_topk_device(const uint32_t* d_rows, const float* d_scores,
                                   const uint32_t* d_list_n, uint32_t n_lists, uint32_t k_in,
                                   uint32_t k_out, uint32_t* d_out_rows, float* d_out_scores,
                                   uint32_t* d_out_n, void* hip_stream) {
    if (!d_rows || !d_scores || !d_out_rows || !d_out_scores || !d_out_n)
        return fail(TRAG_INVALID_ARG, "null argument");
    if ((uint64_t)n_lists * k_in > CAND_CAP)
        return fail(TRAG_UNSUPPORTED, "n_lists * k_in exceeds " + std::to_string(CAND_CAP));
    HIP_TRY(launch_merge_topk(d_rows, d_scores, d_list_n, n_lists, k_in, k_out, d_out_rows,
                              d_out_scores, d_out_n, (hipStream_t)hip_stream));
    return TRAG_OK;
}

trag_status trag_pack_topk_device(const uint32_t* d_rows, const float* d_scores, const uint32_t* d_n,
                                  uint32_t first_global_row, uint32_t n_queries, uint32_t k,
                                  uint32_t* d_packed, void* hip_stream) {
    if (!d_rows || !d_scores || !d_n || !d_packed) return fail(TRAG_INVALID_ARG, "null argument");
    if (n_queries == 0) return TRAG_OK;
    HIP_TRY(launch_pack_topk(d_rows, d_scores, d_n, first_global_row, n_queries, k, d_packed,
                             (hipStream_t)hip_stream));
    return TRAG_OK;
}

trag_status trag_merge_packed_device(const uint32_t* d_gathered, uint32_t n_lists,
                                     uint32_t n_queries, uint32_t k, uint32_t* d_out_rows,
                                     float* d_out_scores, uint32_t* d_out_n, void* hip_stream) {
    if (!d_gathered || !d_out_rows || !d_out_scores || !d_out_n)
        return fail(TRAG_INVALID_ARG, "null argument");
    if ((uint64_t)n_lists * k > CAND_CAP)
        return fail(TRAG_UNSUPPORTED, "n_lists * k exceeds " + std::to_string(CAND_CAP));
    if (n_queries == 0) return TRAG_OK;
    HIP_TRY(launch_merge_packed(d_gathered, n_lists, n_queries, k, d_out_rows, d_out_scores, d_out_n,
                                (hipStream_t)hip_stream));
    return TRAG_OK;
}

trag_status trag_store_last_stats(const trag_store* s, trag_search_stats* out) {
    if (!s || !out) return fail(TRAG_INVALID_ARG, "null argument");
    *out = s->stats;
    return TRAG_OK;
}

}  // extern "C"
