// Row-sharded search over the GPUs of one node, behind the C ABI (include/trag.h, "sharded search").
//
// The corpus is partitioned by rows, one trag_store per GPU; a query batch goes to every shard, each
// shard returns its own top k with GLOBAL row ids, ONE all-gather moves the per-shard lists
// (2k+1 words per query and shard) and every GPU merges them with the ordering rule of a single
// store (score descending, row ascending). The reference is single-process and has no counterpart;
// the call this serves is the dense search of HybridRetriever::retrieve (src/retrieve.rs:181).
//
// Two forms:
//   trag_shard_group_*   one rank = one process = one GPU; ranks meet through an RCCL unique id that
//                        the host program carries from rank 0 to the others (any channel it has)
//   trag_sharded_*       one process drives n GPUs: one communicator (ncclCommInitAll), one stream
//                        per GPU, queries broadcast, local search, all-gather, merge on every GPU
//
// RCCL is resolved at run time (dlopen), not linked: the host process decides which copy lives in
// it. A PyTorch process already holds the wheel's own librccl / libamdhip64 under the same SONAMEs,
// and a second copy bound to a second HIP runtime would not see this library's streams and buffers.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and enums only: every function is looked up with dlsym

#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "trag.h"
#include "trag_internal.h"

using namespace trag;

namespace {

#define SH_TRY(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return set_error(_e == hipErrorOutOfMemory ? TRAG_OOM : TRAG_DEVICE_ERROR,        \
                             std::string(#expr) + ": " + hipGetErrorString(_e));              \
    } while (0)

struct Rccl {
    void* handle = nullptr;
    std::string error;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    // what a communicator says about itself (optional symbols: absent -> trag_shard_group_info says so)
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
};

// Loads RCCL once per process. TRAG_RCCL_LIB names a file to use instead of the search below.
const Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        std::vector<std::string> names;
        if (const char* e = getenv("TRAG_RCCL_LIB")) names.push_back(e);
        // by SONAME first: returns the copy the process already holds, if any
        names.insert(names.end(), {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"});
        for (const std::string& n : names) {
            r.handle = dlopen(n.c_str(), RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
            r.error += n + ": " + (dlerror() ? dlerror() : "not found") + "; ";
        }
        if (!r.handle) return;
        bool ok = true;
        auto sym = [&](const char* name) {
            void* p = dlsym(r.handle, name);
            if (!p) { ok = false; r.error += std::string("missing symbol ") + name + "; "; }
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(sym("ncclGetVersion"));
        if (!ok) { dlclose(r.handle); r.handle = nullptr; return; }
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.handle, "ncclCommCount"));
        r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(dlsym(r.handle, "ncclCommUserRank"));
        r.CommCuDevice = reinterpret_cast<decltype(r.CommCuDevice)>(dlsym(r.handle, "ncclCommCuDevice"));
    });
    return r;
}

trag_status need_rccl(const Rccl** out) {
    const Rccl& r = rccl();
    if (!r.handle) return set_error(TRAG_UNSUPPORTED, "RCCL is not available in this process: " + r.error);
    *out = &r;
    return TRAG_OK;
}

#define NCCL_TRY(R, expr)                                                                     \
    do {                                                                                      \
        ncclResult_t _r = (expr);                                                             \
        if (_r != ncclSuccess)                                                                \
            return set_error(TRAG_DEVICE_ERROR, std::string(#expr) + ": " + (R)->GetErrorString(_r)); \
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

inline size_t rec_words(uint32_t k) { return 2 * (size_t)k + 1; }

// Device buffers of one in-flight exchange: the local search's results, their wire records and
// the gathered records of every shard.
struct ExchangeSlot {
    uint32_t* rows = nullptr;
    float* scores = nullptr;
    uint32_t* n = nullptr;
    uint32_t* status = nullptr;
    uint32_t* pack = nullptr;
    uint32_t* gathered = nullptr;
    hipEvent_t searched = nullptr;  // the local search into this slot has been enqueued up to here
    hipEvent_t done = nullptr;      // pack + gather + merge of this slot have been enqueued up to here
    bool used = false;
};

trag_status slot_alloc(ExchangeSlot& sl, uint32_t world, uint32_t k_max, uint32_t max_queries) {
    const size_t nk = (size_t)max_queries * k_max;
    SH_TRY(hipMalloc(&sl.rows, std::max<size_t>(nk, 1) * sizeof(uint32_t)));
    SH_TRY(hipMalloc(&sl.scores, std::max<size_t>(nk, 1) * sizeof(float)));
    SH_TRY(hipMalloc(&sl.n, (size_t)max_queries * sizeof(uint32_t)));
    SH_TRY(hipMalloc(&sl.status, (size_t)max_queries * sizeof(uint32_t)));
    SH_TRY(hipMalloc(&sl.pack, (size_t)max_queries * rec_words(k_max) * sizeof(uint32_t)));
    SH_TRY(hipMalloc(&sl.gathered, (size_t)world * max_queries * rec_words(k_max) * sizeof(uint32_t)));
    SH_TRY(hipEventCreateWithFlags(&sl.searched, hipEventDisableTiming));
    SH_TRY(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    return TRAG_OK;
}

void slot_free(ExchangeSlot& sl) {
    void* ptrs[] = {sl.rows, sl.scores, sl.n, sl.status, sl.pack, sl.gathered};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (sl.searched) (void)hipEventDestroy(sl.searched);
    if (sl.done) (void)hipEventDestroy(sl.done);
    sl = ExchangeSlot{};
}

}  // namespace

// ---------------------------------------------------------------------------
// one rank per process
// ---------------------------------------------------------------------------
struct trag_shard_group {
    trag_shard_config cfg{};
    int device = 0;
    ncclComm_t comm = nullptr;
    hipStream_t side = nullptr;     // exchange stream (overlap mode) and the host entry's stream
    ExchangeSlot slot[2];
    uint32_t next = 0;
    trag_store* piped = nullptr;    // the local store this group has switched to pipelined mode (overlap)
    // Overlap mode, one query per call: the local store defers its selects (the select of call i rides
    // in the launch of call i + 1's scan), so the records of call i exist only then -- and call i + 1
    // (or trag_shard_fence, or any other call) enqueues call i's exchange on the side stream. The
    // caller's stream then carries one launch and one event record per call. The local results go
    // through a ring of small entries; every RING_WAIT calls the caller's stream waits for an exchange
    // enqueued at least RING_WAIT calls earlier, which keeps an entry from being rewritten before the
    // exchange that read it last has finished (the ring is four such periods long).
    static constexpr uint32_t RING = 64, RING_WAIT = 16;
    uint32_t* ring_rows = nullptr;
    float* ring_scores = nullptr;
    uint32_t* ring_n = nullptr;
    uint32_t* ring_status = nullptr;
    uint32_t* ring_pack = nullptr;       // (one each: the exchanges follow one another on the side stream)
    uint32_t* ring_gathered = nullptr;
    hipEvent_t ev_scan = nullptr, ev_done = nullptr, ev_mark[4] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t ring_calls = 0, ring_exchanged = 0;
    struct PendingExchange {
        bool on = false;
        trag_store* local = nullptr;
        uint32_t entry = 0, k = 0, row0 = 0, row_mul = 1;
        uint32_t* out_rows = nullptr;
        float* out_scores = nullptr;
        uint32_t* out_n = nullptr;
        uint32_t* status = nullptr;
        hipStream_t st = nullptr;
    } pend;
    uint32_t* h_pack = nullptr;     // pinned: records of the host entry, then its merged results
    size_t h_words = 0;
    std::mutex mu;
};

namespace {

trag_status ring_alloc(trag_shard_group* g) {
    if (g->ring_rows) return TRAG_OK;
    const size_t km = g->cfg.k_max, R = trag_shard_group::RING;
    SH_TRY(hipMalloc(&g->ring_rows, R * km * sizeof(uint32_t)));
    SH_TRY(hipMalloc(&g->ring_scores, R * km * sizeof(float)));
    SH_TRY(hipMalloc(&g->ring_n, R * sizeof(uint32_t)));
    SH_TRY(hipMalloc(&g->ring_status, R * sizeof(uint32_t)));
    SH_TRY(hipMalloc(&g->ring_pack, rec_words(g->cfg.k_max) * sizeof(uint32_t)));
    SH_TRY(hipMalloc(&g->ring_gathered, (size_t)g->cfg.world * rec_words(g->cfg.k_max) * sizeof(uint32_t)));
    SH_TRY(hipEventCreateWithFlags(&g->ev_scan, hipEventDisableTiming));
    SH_TRY(hipEventCreateWithFlags(&g->ev_done, hipEventDisableTiming));
    for (hipEvent_t& e : g->ev_mark) SH_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return TRAG_OK;
}

// pack -> all-gather -> merge of one single-query call whose local results are in ring entry P.entry,
// on the side stream (which the caller has ordered behind the select that wrote the entry)
trag_status ring_exchange(trag_shard_group* g, const Rccl* R, const trag_shard_group::PendingExchange& P, bool local_ok) {
    hipStream_t xs = g->side;
    const size_t km = g->cfg.k_max, words = rec_words(P.k);
    if (local_ok)
        SH_TRY(launch_pack_topk(g->ring_rows + P.entry * km, g->ring_scores + P.entry * km, g->ring_n + P.entry, P.row0, 1, P.k,
                                g->ring_pack, xs, P.row_mul ? P.row_mul : 1u, g->ring_status + P.entry));
    else
        SH_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(g->ring_pack), (int)(ST_NEED_REFINE << 16), words, xs));
    NCCL_TRY(R, R->AllGather(g->ring_pack, g->ring_gathered, words, ncclUint32, g->comm, xs));
    SH_TRY(launch_merge_packed(g->ring_gathered, g->cfg.world, 1, P.k, P.out_rows, P.out_scores, P.out_n, xs, P.status));
    SH_TRY(hipEventRecord(g->ev_done, xs));
    const uint64_t e = g->ring_exchanged++;
    if (e % trag_shard_group::RING_WAIT == trag_shard_group::RING_WAIT - 1)
        SH_TRY(hipEventRecord(g->ev_mark[(e / trag_shard_group::RING_WAIT) % 4], xs));
    return TRAG_OK;
}

// The exchange a deferred single-query call still owes: its select is launched (the local store's fence),
// the side stream ordered behind it, the exchange enqueued.
trag_status flush_pending_exchange(trag_shard_group* g, const Rccl* R) {
    if (!g->pend.on) return TRAG_OK;
    const trag_shard_group::PendingExchange P = g->pend;
    g->pend.on = false;
    trag_status ts = trag_store_fence(P.local, P.st);
    if (ts != TRAG_OK) return ts;
    SH_TRY(hipEventRecord(g->ev_scan, P.st));
    SH_TRY(hipStreamWaitEvent(g->side, g->ev_scan, 0));
    return ring_exchange(g, R, P, true);
}

}  // namespace

extern "C" {

trag_status trag_shard_unique_id(uint8_t* id) {
    if (!id) return set_error(TRAG_INVALID_ARG, "null argument");
    const Rccl* R = nullptr;
    trag_status ts = need_rccl(&R);
    if (ts != TRAG_OK) return ts;
    static_assert(sizeof(ncclUniqueId) == TRAG_SHARD_ID_BYTES, "unique id size");
    ncclUniqueId u;
    NCCL_TRY(R, R->GetUniqueId(&u));
    std::memcpy(id, &u, sizeof(u));
    return TRAG_OK;
}

void trag_shard_group_destroy(trag_shard_group* g) {
    if (!g) return;
    {
        DeviceScope scope(g->device);
        if (g->side) (void)hipStreamSynchronize(g->side);
        (void)hipDeviceSynchronize();
        if (g->comm && rccl().handle) (void)rccl().CommDestroy(g->comm);
        slot_free(g->slot[0]);
        slot_free(g->slot[1]);
        void* ring[] = {g->ring_rows, g->ring_scores, g->ring_n, g->ring_status, g->ring_pack, g->ring_gathered};
        for (void* p : ring)
            if (p) (void)hipFree(p);
        hipEvent_t evs[] = {g->ev_scan, g->ev_done, g->ev_mark[0], g->ev_mark[1], g->ev_mark[2], g->ev_mark[3]};
        for (hipEvent_t e : evs)
            if (e) (void)hipEventDestroy(e);
        if (g->h_pack) (void)hipHostFree(g->h_pack);
        if (g->side) (void)hipStreamDestroy(g->side);
    }
    delete g;
}

trag_status trag_shard_group_create(const trag_shard_config* cfg, const uint8_t* id, trag_shard_group** out) {
    if (!cfg || !id || !out) return set_error(TRAG_INVALID_ARG, "null argument");
    *out = nullptr;
    if (cfg->world == 0 || cfg->rank >= cfg->world) return set_error(TRAG_INVALID_CONFIG, "rank must be below world");
    if (cfg->k_max == 0 || cfg->max_queries == 0) return set_error(TRAG_INVALID_CONFIG, "k_max and max_queries must be positive");
    if ((uint64_t)cfg->world * cfg->k_max > CAND_CAP)
        return set_error(TRAG_UNSUPPORTED, "world * k_max exceeds " + std::to_string(CAND_CAP) + " (the merge sorts in one workgroup)");
    const Rccl* R = nullptr;
    trag_status ts = need_rccl(&R);
    if (ts != TRAG_OK) return ts;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return set_error(TRAG_DEVICE_ERROR, "no HIP device available (this library has no CPU path)");
    int dev = cfg->device;
    if (dev < 0) SH_TRY(hipGetDevice(&dev));
    if (dev >= ndev) return set_error(TRAG_INVALID_CONFIG, "device ordinal out of range");
    DeviceScope scope(dev);
    trag_shard_group* g = new trag_shard_group();
    g->cfg = *cfg;
    g->cfg.device = dev;
    g->device = dev;
    auto bail = [&](trag_status st) { trag_shard_group_destroy(g); return st; };
    hipError_t e = hipStreamCreateWithFlags(&g->side, hipStreamNonBlocking);
    if (e != hipSuccess) return bail(set_error(TRAG_DEVICE_ERROR, std::string("hipStreamCreate: ") + hipGetErrorString(e)));
    for (ExchangeSlot& sl : g->slot) {
        ts = slot_alloc(sl, cfg->world, cfg->k_max, cfg->max_queries);
        if (ts != TRAG_OK) return bail(ts);
    }
    g->h_words = (size_t)cfg->max_queries * rec_words(cfg->k_max) + cfg->max_queries;
    e = hipHostMalloc(reinterpret_cast<void**>(&g->h_pack), g->h_words * sizeof(uint32_t), hipHostMallocDefault);
    if (e != hipSuccess) return bail(set_error(TRAG_OOM, std::string("hipHostMalloc: ") + hipGetErrorString(e)));
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    ncclResult_t r = R->CommInitRank(&g->comm, (int)cfg->world, u, (int)cfg->rank);
    if (r != ncclSuccess) {
        g->comm = nullptr;
        return bail(set_error(TRAG_DEVICE_ERROR, std::string("ncclCommInitRank: ") + R->GetErrorString(r)));
    }
    *out = g;
    return TRAG_OK;
}

trag_status trag_shard_fence(trag_shard_group* g, void* hip_stream) {
    if (!g) return set_error(TRAG_INVALID_ARG, "null group");
    std::lock_guard<std::mutex> lk(g->mu);
    DeviceScope scope(g->device);
    if (g->pend.on) {
        const Rccl* R = nullptr;
        trag_status ts = need_rccl(&R);
        if (ts == TRAG_OK) ts = flush_pending_exchange(g, R);
        if (ts != TRAG_OK) return ts;
    }
    for (ExchangeSlot& sl : g->slot)
        if (sl.used) SH_TRY(hipStreamWaitEvent((hipStream_t)hip_stream, sl.done, 0));
    if (g->ring_exchanged) SH_TRY(hipStreamWaitEvent((hipStream_t)hip_stream, g->ev_done, 0));
    return TRAG_OK;
}

trag_status trag_shard_search_device(trag_shard_group* g, trag_store* local, uint32_t row0, uint32_t row_mul,
                                     const float* d_queries, uint32_t n_queries, uint32_t dim, uint32_t k,
                                     uint32_t* d_out_rows, float* d_out_scores, uint32_t* d_out_n,
                                     uint32_t* d_status, void* hip_stream) {
    if (!g || !local) return set_error(TRAG_INVALID_ARG, "null argument");
    if (n_queries == 0) return TRAG_OK;
    if (!d_queries || !d_out_n || (k && (!d_out_rows || !d_out_scores))) return set_error(TRAG_INVALID_ARG, "null argument");
    if (n_queries > g->cfg.max_queries || k > g->cfg.k_max)
        return set_error(TRAG_INVALID_ARG, "batch or k above what the group was created for (max_queries " +
                                               std::to_string(g->cfg.max_queries) + ", k_max " + std::to_string(g->cfg.k_max) + ")");
    trag_config lc{};
    trag_status ts = trag_store_config(local, &lc);
    if (ts != TRAG_OK) return ts;
    if (lc.device != g->device) return set_error(TRAG_INVALID_ARG, "the local store lives on another device than the group");
    const Rccl* R = nullptr;
    ts = need_rccl(&R);
    if (ts != TRAG_OK) return ts;
    std::lock_guard<std::mutex> lk(g->mu);
    DeviceScope scope(g->device);
    hipStream_t st = (hipStream_t)hip_stream;
    if (k == 0) {
        SH_TRY(hipMemsetAsync(d_out_n, 0, n_queries * sizeof(uint32_t), st));
        if (d_status) SH_TRY(hipMemsetAsync(d_status, 0, n_queries * sizeof(uint32_t), st));
        return TRAG_OK;
    }
    // overlap mode: the select of a single-query step leaves the critical path of the stream the scans
    // run on (the local store defers it into its next scan's launch)
    if (g->cfg.overlap && g->piped != local) {
        trag_status tp = trag_store_set_pipelined(local, 2);
        if (tp != TRAG_OK) return tp;
        g->piped = local;
    }
    if (g->cfg.overlap && n_queries == 1) {
        ts = ring_alloc(g);
        if (ts != TRAG_OK) return ts;
        const uint64_t c = g->ring_calls++;
        constexpr uint64_t W = trag_shard_group::RING_WAIT;
        if (c % W == 0 && c >= 2 * W) SH_TRY(hipStreamWaitEvent(st, g->ev_mark[(c / W - 2) % 4], 0));
        trag_shard_group::PendingExchange me;
        me.local = local;
        me.entry = (uint32_t)(c % trag_shard_group::RING);
        me.k = k;
        me.row0 = row0;
        me.row_mul = row_mul;
        me.out_rows = d_out_rows;
        me.out_scores = d_out_scores;
        me.out_n = d_out_n;
        me.status = d_status;
        me.st = st;
        const size_t km = g->cfg.k_max;
        // (a pending exchange of another stream's call: settled first, so that this call's launch need not)
        if (g->pend.on && (g->pend.st != st || g->pend.local != local)) {
            ts = flush_pending_exchange(g, R);
            if (ts != TRAG_OK) return ts;
        }
        const trag_status local_ts = trag_store_search_device(local, d_queries, 1, dim, k, g->ring_rows + me.entry * km,
                                                              g->ring_scores + me.entry * km, g->ring_n + me.entry,
                                                              g->ring_status + me.entry, st);
        const std::string local_msg = local_ts != TRAG_OK ? std::string(trag_last_error()) : std::string();
        if (local_ts != TRAG_OK && g->pend.on) (void)trag_store_fence(local, st);   // (the select the pending exchange waits for)
        const bool deferred_now = local_ts == TRAG_OK && store_select_pending(local);
        // whatever was pending before this call has its select on `st` by now: it rode in this call's
        // launch, or the store launched it on its own
        if (g->pend.on || !deferred_now) {
            SH_TRY(hipEventRecord(g->ev_scan, st));
            SH_TRY(hipStreamWaitEvent(g->side, g->ev_scan, 0));
            // (a route whose last stages the store runs on its own side stream -- the mirror scan's, the
            // multi-level select's -- has its results there, not on `st`)
            if (!deferred_now && local_ts == TRAG_OK) {
                ts = store_wait_side(local, g->side);
                if (ts != TRAG_OK) return ts;
            }
        }
        if (g->pend.on) {
            const trag_shard_group::PendingExchange P = g->pend;
            g->pend.on = false;
            ts = ring_exchange(g, R, P, true);
            if (ts != TRAG_OK) return ts;
        }
        if (deferred_now) {
            g->pend = me;
            g->pend.on = true;
            return TRAG_OK;
        }
        ts = ring_exchange(g, R, me, local_ts == TRAG_OK);
        if (ts != TRAG_OK) return ts;
        if (local_ts != TRAG_OK) return set_error(local_ts, "local search of this rank failed (the exchange was still joined): " + local_msg);
        return TRAG_OK;
    }
    ts = flush_pending_exchange(g, R);
    if (ts != TRAG_OK) return ts;
    ExchangeSlot& sl = g->slot[g->next];
    g->next ^= 1u;
    // the exchange that used this slot two calls ago may still be reading it on the side stream
    if (sl.used) SH_TRY(hipStreamWaitEvent(st, sl.done, 0));
    // A rank whose local search cannot be enqueued still takes part in the collective -- its peers have
    // entered it or are about to, and an all-gather one rank never joins hangs them all: it sends
    // records that hold no row and carry status 3, so every rank sees the query as unfinished, and
    // this rank returns its error once the exchange is enqueued.
    const trag_status local_ts =
        trag_store_search_device(local, d_queries, n_queries, dim, k, sl.rows, sl.scores, sl.n, sl.status, st);
    const std::string local_msg = local_ts != TRAG_OK ? std::string(trag_last_error()) : std::string();
    hipStream_t xs = st;
    if (g->cfg.overlap) {
        SH_TRY(hipEventRecord(sl.searched, st));
        SH_TRY(hipStreamWaitEvent(g->side, sl.searched, 0));
        xs = g->side;
    }
    {
        // (a pipelined store -- the group's own doing in overlap mode, or the caller's -- has this call's last
        // stages on its own side stream, not on st; a no-op otherwise)
        trag_status tf = trag_store_fence(local, xs);
        if (tf != TRAG_OK) return tf;
    }
    const size_t words = (size_t)n_queries * rec_words(k);
    if (local_ts == TRAG_OK)
        SH_TRY(launch_pack_topk(sl.rows, sl.scores, sl.n, row0, n_queries, k, sl.pack, xs, row_mul ? row_mul : 1u, sl.status));
    else   // every word (ST_NEED_REFINE << 16) | 0: read as a count word, no rows and that status
        SH_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(sl.pack), (int)(ST_NEED_REFINE << 16), words, xs));
    NCCL_TRY(R, R->AllGather(sl.pack, sl.gathered, words, ncclUint32, g->comm, xs));
    SH_TRY(launch_merge_packed(sl.gathered, g->cfg.world, n_queries, k, d_out_rows, d_out_scores, d_out_n, xs, d_status));
    SH_TRY(hipEventRecord(sl.done, xs));
    sl.used = true;
    if (local_ts != TRAG_OK) return set_error(local_ts, "local search of this rank failed (the exchange was still joined): " + local_msg);
    return TRAG_OK;
}

trag_status trag_shard_group_info(trag_shard_group* g, uint32_t* comm_world, uint32_t* comm_rank, int32_t* comm_device) {
    if (!g) return set_error(TRAG_INVALID_ARG, "null group");
    const Rccl* R = nullptr;
    trag_status ts = need_rccl(&R);
    if (ts != TRAG_OK) return ts;
    if (!R->CommCount || !R->CommUserRank || !R->CommCuDevice)
        return set_error(TRAG_UNSUPPORTED, "this RCCL exports no ncclCommCount / ncclCommUserRank / ncclCommCuDevice");
    std::lock_guard<std::mutex> lk(g->mu);
    int n = 0, r = 0, d = 0;
    NCCL_TRY(R, R->CommCount(g->comm, &n));
    NCCL_TRY(R, R->CommUserRank(g->comm, &r));
    NCCL_TRY(R, R->CommCuDevice(g->comm, &d));
    if (comm_world) *comm_world = (uint32_t)n;
    if (comm_rank) *comm_rank = (uint32_t)r;
    if (comm_device) *comm_device = d;
    return TRAG_OK;
}

trag_status trag_shard_search_batch(trag_shard_group* g, trag_store* local, uint32_t row0, uint32_t row_mul,
                                    const float* queries, uint32_t n_queries, uint32_t dim, uint32_t k,
                                    uint32_t* out_rows, float* out_scores, uint32_t* out_n) {
    if (!g || !local) return set_error(TRAG_INVALID_ARG, "null argument");
    if (n_queries == 0) return TRAG_OK;
    if (!queries || !out_n || (k && (!out_rows || !out_scores))) return set_error(TRAG_INVALID_ARG, "null argument");
    if (k > g->cfg.k_max) return set_error(TRAG_INVALID_ARG, "k above the group's k_max");
    const Rccl* R = nullptr;
    trag_status ts = need_rccl(&R);
    if (ts != TRAG_OK) return ts;
    if (k == 0) {
        for (uint32_t b = 0; b < n_queries; ++b) out_n[b] = 0;
        return TRAG_OK;
    }
    if (row_mul == 0) row_mul = 1;
    std::lock_guard<std::mutex> lk(g->mu);
    DeviceScope scope(g->device);
    ts = flush_pending_exchange(g, R);
    if (ts != TRAG_OK) return ts;
    for (ExchangeSlot& sl : g->slot)
        if (sl.used) SH_TRY(hipEventSynchronize(sl.done));
    if (g->ring_exchanged) SH_TRY(hipEventSynchronize(g->ev_done));
    ExchangeSlot& sl = g->slot[0];
    ExchangeSlot& res = g->slot[1];  // its result arrays receive the merged lists
    hipStream_t st = g->side;
    const size_t rec = rec_words(k);
    std::vector<uint32_t> l_rows((size_t)g->cfg.max_queries * k), l_n(g->cfg.max_queries);
    std::vector<float> l_scores((size_t)g->cfg.max_queries * k);
    for (uint32_t b0 = 0; b0 < n_queries; b0 += g->cfg.max_queries) {
        const uint32_t nb = std::min<uint32_t>(g->cfg.max_queries, n_queries - b0);
        // the local store's host entry finishes every query itself (fallbacks included)
        ts = trag_store_search_batch(local, queries + (size_t)b0 * dim, nb, dim, k, l_rows.data(), l_scores.data(), l_n.data());
        if (ts != TRAG_OK) return ts;
        for (uint32_t b = 0; b < nb; ++b) {
            uint32_t* r = g->h_pack + (size_t)b * rec;
            const uint32_t cnt = std::min(l_n[b], k);
            for (uint32_t j = 0; j < k; ++j) {
                uint32_t bits = 0, row = 0;
                if (j < cnt) {
                    std::memcpy(&bits, &l_scores[(size_t)b * k + j], 4);
                    row = l_rows[(size_t)b * k + j] * row_mul + row0;
                }
                r[j] = bits;
                r[k + j] = row;
            }
            r[2 * (size_t)k] = cnt;
        }
        const size_t words = (size_t)nb * rec;
        SH_TRY(hipMemcpyAsync(sl.pack, g->h_pack, words * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        NCCL_TRY(R, R->AllGather(sl.pack, sl.gathered, words, ncclUint32, g->comm, st));
        SH_TRY(launch_merge_packed(sl.gathered, g->cfg.world, nb, k, res.rows, res.scores, res.n, st, nullptr));
        uint32_t* h = g->h_pack;
        const size_t nk = (size_t)nb * k;
        SH_TRY(hipMemcpyAsync(h, res.rows, nk * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        SH_TRY(hipStreamSynchronize(st));
        std::memcpy(out_rows + (size_t)b0 * k, h, nk * sizeof(uint32_t));
        SH_TRY(hipMemcpyAsync(h, res.scores, nk * sizeof(float), hipMemcpyDeviceToHost, st));
        SH_TRY(hipMemcpyAsync(h + nk, res.n, nb * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        SH_TRY(hipStreamSynchronize(st));
        std::memcpy(out_scores + (size_t)b0 * k, h, nk * sizeof(float));
        std::memcpy(out_n + b0, h + nk, nb * sizeof(uint32_t));
    }
    return TRAG_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// one process, n GPUs
// ---------------------------------------------------------------------------
struct ShardDevice {
    int device = 0;
    trag_store* store = nullptr;
    hipStream_t stream = nullptr;
    ncclComm_t comm = nullptr;
    float* d_q = nullptr;
    ExchangeSlot slot;              // local results, records, gathered records
    uint32_t* out_rows = nullptr;   // merged results
    float* out_scores = nullptr;
    uint32_t* out_n = nullptr;
    uint32_t* out_status = nullptr;
    hipEvent_t packed = nullptr;    // this shard's records are ready (copy exchange)
};

struct trag_sharded_store {
    trag_config cfg{};
    uint32_t G = 0;
    bool use_rccl = true;
    std::vector<ShardDevice> dev;
    uint64_t n_rows = 0;            // global rows ever inserted: global row g = shard g % G, local row g / G
    uint32_t k_cap = 0, q_cap = 0;  // what the exchange buffers were sized for
    float* h_q = nullptr;           // pinned staging of the queries
    uint32_t* h_out = nullptr;      // pinned: rows | scores | counts | statuses of device 0
    std::mutex mu;
    uint64_t fallbacks = 0;         // queries finished through the shards' host entries
};

namespace {

void sharded_free_buffers(trag_sharded_store* s) {
    for (ShardDevice& d : s->dev) {
        DeviceScope scope(d.device);
        if (d.stream) (void)hipStreamSynchronize(d.stream);
        slot_free(d.slot);
        void* ptrs[] = {d.d_q, d.out_rows, d.out_scores, d.out_n, d.out_status};
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
        d.d_q = nullptr;
        d.out_rows = d.out_n = d.out_status = nullptr;
        d.out_scores = nullptr;
    }
    if (s->h_q) (void)hipHostFree(s->h_q);
    if (s->h_out) (void)hipHostFree(s->h_out);
    s->h_q = nullptr;
    s->h_out = nullptr;
    s->k_cap = s->q_cap = 0;
}

trag_status sharded_ensure(trag_sharded_store* s, uint32_t nq, uint32_t k) {
    if (nq <= s->q_cap && k <= s->k_cap) return TRAG_OK;
    const uint32_t qc = std::max(nq, s->q_cap), kc = std::max(k, s->k_cap);
    sharded_free_buffers(s);
    for (ShardDevice& d : s->dev) {
        DeviceScope scope(d.device);
        trag_status ts = slot_alloc(d.slot, s->G, kc, qc);
        if (ts != TRAG_OK) return ts;
        const size_t nk = (size_t)qc * kc;
        SH_TRY(hipMalloc(&d.d_q, (size_t)qc * s->cfg.dim * sizeof(float)));
        SH_TRY(hipMalloc(&d.out_rows, nk * sizeof(uint32_t)));
        SH_TRY(hipMalloc(&d.out_scores, nk * sizeof(float)));
        SH_TRY(hipMalloc(&d.out_n, (size_t)qc * sizeof(uint32_t)));
        SH_TRY(hipMalloc(&d.out_status, (size_t)qc * sizeof(uint32_t)));
    }
    SH_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_q), (size_t)qc * s->cfg.dim * sizeof(float), hipHostMallocDefault));
    SH_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_out), (2 * (size_t)qc * kc + 2 * (size_t)qc) * sizeof(uint32_t),
                         hipHostMallocDefault));
    s->q_cap = qc;
    s->k_cap = kc;
    return TRAG_OK;
}

// first index i of a block appended at global row g0 that lands on shard j, and how many of the
// block's n rows do
inline void stripe(uint64_t g0, uint64_t n, uint32_t G, uint32_t j, uint64_t* first, uint64_t* count) {
    const uint64_t i0 = (j + G - g0 % G) % G;
    *first = i0;
    *count = i0 < n ? (n - i0 + G - 1) / G : 0;
}

}  // namespace

extern "C" {

void trag_sharded_destroy(trag_sharded_store* s) {
    if (!s) return;
    sharded_free_buffers(s);
    for (ShardDevice& d : s->dev) {
        DeviceScope scope(d.device);
        if (d.comm && rccl().handle) (void)rccl().CommDestroy(d.comm);
        if (d.store) trag_store_destroy(d.store);
        if (d.packed) (void)hipEventDestroy(d.packed);
        if (d.stream) (void)hipStreamDestroy(d.stream);
    }
    delete s;
}

trag_status trag_sharded_create(const trag_config* cfg, uint32_t n_devices, const int32_t* device_ids,
                                uint32_t exchange, trag_sharded_store** out) {
    if (!cfg || !out || (n_devices && !device_ids)) return set_error(TRAG_INVALID_ARG, "null argument");
    *out = nullptr;
    if (n_devices == 0 || n_devices > 64) return set_error(TRAG_INVALID_CONFIG, "n_devices must be 1..64");
    if (exchange > TRAG_EXCHANGE_COPY) return set_error(TRAG_INVALID_CONFIG, "unknown exchange");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return set_error(TRAG_DEVICE_ERROR, "no HIP device available (this library has no CPU path)");
    for (uint32_t j = 0; j < n_devices; ++j)
        if (device_ids[j] < 0 || device_ids[j] >= ndev) return set_error(TRAG_INVALID_CONFIG, "device ordinal out of range");
    trag_sharded_store* s = new trag_sharded_store();
    s->cfg = *cfg;
    s->G = n_devices;
    s->use_rccl = exchange == TRAG_EXCHANGE_RCCL;
    s->dev.resize(n_devices);
    auto bail = [&](trag_status st) { trag_sharded_destroy(s); return st; };
    for (uint32_t j = 0; j < n_devices; ++j) {
        ShardDevice& d = s->dev[j];
        d.device = device_ids[j];
        DeviceScope scope(d.device);
        trag_config c = *cfg;
        c.device = d.device;
        c.reserve_rows = (cfg->reserve_rows + n_devices - 1) / n_devices;
        trag_status ts = trag_store_create(&c, &d.store);
        if (ts != TRAG_OK) return bail(ts);
        hipError_t e = hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&d.packed, hipEventDisableTiming);
        if (e != hipSuccess) return bail(set_error(TRAG_DEVICE_ERROR, std::string("stream/event: ") + hipGetErrorString(e)));
        if (!s->use_rccl)   // direct peer copies where the hardware allows them (best effort)
            for (uint32_t o = 0; o < n_devices; ++o)
                if (device_ids[o] != d.device) {
                    (void)hipDeviceEnablePeerAccess(device_ids[o], 0);
                    (void)hipGetLastError();
                }
    }
    if (s->use_rccl) {
        const Rccl* R = nullptr;
        trag_status ts = need_rccl(&R);
        if (ts != TRAG_OK) return bail(ts);
        std::vector<ncclComm_t> comms(n_devices, nullptr);
        std::vector<int> ids(device_ids, device_ids + n_devices);
        ncclResult_t r = R->CommInitAll(comms.data(), (int)n_devices, ids.data());
        if (r != ncclSuccess)
            return bail(set_error(TRAG_DEVICE_ERROR, std::string("ncclCommInitAll: ") + R->GetErrorString(r)));
        for (uint32_t j = 0; j < n_devices; ++j) s->dev[j].comm = comms[j];
    }
    *out = s;
    return TRAG_OK;
}

uint32_t trag_sharded_devices(const trag_sharded_store* s) { return s ? s->G : 0; }

trag_store* trag_sharded_shard(trag_sharded_store* s, uint32_t i) { return (s && i < s->G) ? s->dev[i].store : nullptr; }

uint64_t trag_sharded_len(const trag_sharded_store* s) {
    uint64_t n = 0;
    if (s)
        for (const ShardDevice& d : s->dev) n += trag_store_len(d.store);
    return n;
}

uint64_t trag_sharded_rows(const trag_sharded_store* s) { return s ? s->n_rows : 0; }

trag_status trag_sharded_insert(trag_sharded_store* s, const float* rows, uint64_t n, uint32_t dim,
                                uint32_t* out_first_row) {
    if (!s) return set_error(TRAG_INVALID_ARG, "null store");
    if (n > 0 && !rows) return set_error(TRAG_INVALID_CONFIG, "chunk must have embedding");
    if (dim != s->cfg.dim)
        return set_error(TRAG_DIM_MISMATCH, "dimension mismatch: expected " + std::to_string(s->cfg.dim) +
                                                ", got " + std::to_string(dim));
    std::lock_guard<std::mutex> lk(s->mu);
    if (s->n_rows + n > 0xFFFFFFF0ull) return set_error(TRAG_INVALID_ARG, "row count exceeds u32 row ids");
    if (out_first_row) *out_first_row = (uint32_t)s->n_rows;
    if (n == 0) return TRAG_OK;
    // room on every shard first: a shard that cannot grow (out of memory on one GPU) must fail the call
    // before any other shard has appended its stripe, or the shards would be out of step for good
    for (uint32_t j = 0; j < s->G; ++j) {
        uint64_t i0 = 0, cnt = 0;
        stripe(s->n_rows, n, s->G, j, &i0, &cnt);
        if (cnt == 0) continue;
        trag_status tr = trag_store_reserve(s->dev[j].store, trag_store_rows(s->dev[j].store) + cnt);
        if (tr != TRAG_OK) return tr;
    }
    std::vector<float> mine;
    for (uint32_t j = 0; j < s->G; ++j) {
        uint64_t i0 = 0, cnt = 0;
        stripe(s->n_rows, n, s->G, j, &i0, &cnt);
        if (cnt == 0) continue;
        const float* src = rows + i0 * dim;
        if (s->G > 1) {
            mine.resize((size_t)cnt * dim);
            for (uint64_t t = 0; t < cnt; ++t)
                std::memcpy(mine.data() + t * dim, rows + (i0 + t * s->G) * dim, (size_t)dim * sizeof(float));
            src = mine.data();
        }
        uint32_t first = 0;
        trag_status ts = trag_store_insert(s->dev[j].store, src, cnt, dim, &first);
        if (ts != TRAG_OK) return ts;
        if (first != (s->n_rows + i0) / s->G) return set_error(TRAG_DEVICE_ERROR, "shard out of step with the striping");
    }
    s->n_rows += n;
    return TRAG_OK;
}

trag_status trag_sharded_fill_synthetic(trag_sharded_store* s, uint64_t n, uint64_t seed, uint32_t* out_first_row) {
    if (!s) return set_error(TRAG_INVALID_ARG, "null store");
    std::lock_guard<std::mutex> lk(s->mu);
    if (s->n_rows + n > 0xFFFFFFF0ull) return set_error(TRAG_INVALID_ARG, "row count exceeds u32 row ids");
    if (out_first_row) *out_first_row = (uint32_t)s->n_rows;
    for (uint32_t j = 0; j < s->G; ++j) {
        uint64_t i0 = 0, cnt = 0;
        stripe(s->n_rows, n, s->G, j, &i0, &cnt);
        if (cnt == 0) continue;
        trag_status ts = store_fill_synthetic(s->dev[j].store, cnt, seed, s->n_rows + i0, s->G, nullptr);
        if (ts != TRAG_OK) return ts;
    }
    s->n_rows += n;
    return TRAG_OK;
}

trag_status trag_sharded_overwrite(trag_sharded_store* s, uint32_t row, const float* vec, uint32_t dim) {
    if (!s) return set_error(TRAG_INVALID_ARG, "null store");
    std::lock_guard<std::mutex> lk(s->mu);
    if (row >= s->n_rows) return set_error(TRAG_INVALID_ARG, "row out of range");
    return trag_store_overwrite(s->dev[row % s->G].store, row / s->G, vec, dim);
}

trag_status trag_sharded_remove(trag_sharded_store* s, uint32_t row) {
    if (!s) return set_error(TRAG_INVALID_ARG, "null store");
    std::lock_guard<std::mutex> lk(s->mu);
    if (row >= s->n_rows) return set_error(TRAG_INVALID_ARG, "row out of range");
    return trag_store_remove(s->dev[row % s->G].store, row / s->G);
}

trag_status trag_sharded_search_batch(trag_sharded_store* s, const float* queries, uint32_t n_queries,
                                      uint32_t dim, uint32_t k, uint32_t* out_rows, float* out_scores,
                                      uint32_t* out_n) {
    if (!s) return set_error(TRAG_INVALID_ARG, "null store");
    if (dim != s->cfg.dim)
        return set_error(TRAG_DIM_MISMATCH, "dimension mismatch: expected " + std::to_string(s->cfg.dim) +
                                                ", got " + std::to_string(dim));
    if (n_queries == 0) return TRAG_OK;
    if (!queries || !out_n || (k && (!out_rows || !out_scores))) return set_error(TRAG_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    const uint64_t live = trag_sharded_len(s);
    if (k == 0 || live == 0) {
        for (uint32_t b = 0; b < n_queries; ++b) out_n[b] = 0;
        return TRAG_OK;
    }
    const uint32_t k_user = k;
    k = (uint32_t)std::min<uint64_t>(k, live);   // any k is legal, buffers follow what can come back
    if ((uint64_t)s->G * k > CAND_CAP)
        return set_error(TRAG_UNSUPPORTED, "n_devices * k exceeds " + std::to_string(CAND_CAP) + " (the merge sorts in one workgroup)");
    const Rccl* R = nullptr;
    if (s->use_rccl) {
        trag_status ts = need_rccl(&R);
        if (ts != TRAG_OK) return ts;
    }
    const uint32_t chunk = std::min<uint32_t>(n_queries, 4096);
    trag_status ts = sharded_ensure(s, chunk, k);
    if (ts != TRAG_OK) return ts;
    const size_t rec = rec_words(k);
    for (uint32_t b0 = 0; b0 < n_queries; b0 += chunk) {
        const uint32_t nb = std::min(chunk, n_queries - b0);
        const size_t q_bytes = (size_t)nb * dim * sizeof(float);
        const size_t words = (size_t)nb * rec;
        std::memcpy(s->h_q, queries + (size_t)b0 * dim, q_bytes);
        // queries: one upload, then a broadcast over the links (RCCL); or one upload per GPU
        if (s->use_rccl) {
            {
                DeviceScope scope(s->dev[0].device);
                SH_TRY(hipMemcpyAsync(s->dev[0].d_q, s->h_q, q_bytes, hipMemcpyHostToDevice, s->dev[0].stream));
            }
            if (s->G > 1) {
                NCCL_TRY(R, R->GroupStart());
                for (ShardDevice& d : s->dev) {
                    ncclResult_t r = R->Broadcast(d.d_q, d.d_q, (size_t)nb * dim, ncclFloat, 0, d.comm, d.stream);
                    if (r != ncclSuccess) {
                        (void)R->GroupEnd();
                        return set_error(TRAG_DEVICE_ERROR, std::string("ncclBroadcast: ") + R->GetErrorString(r));
                    }
                }
                NCCL_TRY(R, R->GroupEnd());
            }
        } else {
            for (ShardDevice& d : s->dev) {
                DeviceScope scope(d.device);
                SH_TRY(hipMemcpyAsync(d.d_q, s->h_q, q_bytes, hipMemcpyHostToDevice, d.stream));
            }
        }
        // local search + wire records, every GPU on its own stream
        for (uint32_t j = 0; j < s->G; ++j) {
            ShardDevice& d = s->dev[j];
            DeviceScope scope(d.device);
            ts = trag_store_search_device(d.store, d.d_q, nb, dim, k, d.slot.rows, d.slot.scores, d.slot.n,
                                          d.slot.status, d.stream);
            if (ts != TRAG_OK) return ts;
            SH_TRY(launch_pack_topk(d.slot.rows, d.slot.scores, d.slot.n, j, nb, k, d.slot.pack, d.stream, s->G,
                                    d.slot.status));
            if (!s->use_rccl) SH_TRY(hipEventRecord(d.packed, d.stream));
        }
        // the exchange: ONE all-gather of the records
        if (s->use_rccl) {
            NCCL_TRY(R, R->GroupStart());
            for (ShardDevice& d : s->dev) {
                ncclResult_t r = R->AllGather(d.slot.pack, d.slot.gathered, words, ncclUint32, d.comm, d.stream);
                if (r != ncclSuccess) {
                    (void)R->GroupEnd();
                    return set_error(TRAG_DEVICE_ERROR, std::string("ncclAllGather: ") + R->GetErrorString(r));
                }
            }
            NCCL_TRY(R, R->GroupEnd());
        } else {
            for (ShardDevice& dst : s->dev) {
                DeviceScope scope(dst.device);
                for (uint32_t j = 0; j < s->G; ++j) {
                    ShardDevice& src = s->dev[j];
                    if (&src != &dst) SH_TRY(hipStreamWaitEvent(dst.stream, src.packed, 0));
                    SH_TRY(hipMemcpyAsync(dst.slot.gathered + (size_t)j * words, src.slot.pack, words * sizeof(uint32_t),
                                          hipMemcpyDefault, dst.stream));
                }
            }
        }
        // merge on every GPU (each then holds the global result); GPU 0's copy goes back
        for (ShardDevice& d : s->dev) {
            DeviceScope scope(d.device);
            SH_TRY(launch_merge_packed(d.slot.gathered, s->G, nb, k, d.out_rows, d.out_scores, d.out_n, d.stream,
                                       d.out_status));
        }
        const size_t nk = (size_t)nb * k;
        {
            ShardDevice& d = s->dev[0];
            DeviceScope scope(d.device);
            SH_TRY(hipMemcpyAsync(s->h_out, d.out_rows, nk * sizeof(uint32_t), hipMemcpyDeviceToHost, d.stream));
            SH_TRY(hipMemcpyAsync(s->h_out + nk, d.out_scores, nk * sizeof(float), hipMemcpyDeviceToHost, d.stream));
            SH_TRY(hipMemcpyAsync(s->h_out + 2 * nk, d.out_n, nb * sizeof(uint32_t), hipMemcpyDeviceToHost, d.stream));
            SH_TRY(hipMemcpyAsync(s->h_out + 2 * nk + nb, d.out_status, nb * sizeof(uint32_t), hipMemcpyDeviceToHost, d.stream));
        }
        // every stream is drained: the next chunk (or call) reuses the buffers the copies read
        for (ShardDevice& d : s->dev) {
            DeviceScope scope(d.device);
            SH_TRY(hipStreamSynchronize(d.stream));
        }
        const uint32_t* h_rows = s->h_out;
        const uint32_t* h_scores = s->h_out + nk;
        const uint32_t* h_n = s->h_out + 2 * nk;
        const uint32_t* h_status = h_n + nb;
        for (uint32_t b = 0; b < nb; ++b) {
            uint32_t* o_rows = out_rows + (size_t)(b0 + b) * k_user;
            float* o_scores = out_scores + (size_t)(b0 + b) * k_user;
            if (h_status[b] == ST_OK) {
                std::memcpy(o_rows, h_rows + (size_t)b * k, k * sizeof(uint32_t));
                std::memcpy(o_scores, h_scores + (size_t)b * k, k * sizeof(float));
                out_n[b0 + b] = h_n[b];
                continue;
            }
            // some shard's fast path gave this query up: finish it through the shards' host entries
            // (they repeat with the plan that works and remember it) and merge on the host
            s->fallbacks += 1;
            std::vector<uint32_t> l_rows((size_t)s->G * k), l_n(s->G);
            std::vector<float> l_scores((size_t)s->G * k);
            for (uint32_t j = 0; j < s->G; ++j) {
                ts = trag_store_search(s->dev[j].store, queries + (size_t)(b0 + b) * dim, dim, k,
                                       l_rows.data() + (size_t)j * k, l_scores.data() + (size_t)j * k, &l_n[j]);
                if (ts != TRAG_OK) return ts;
                for (uint32_t t = 0; t < l_n[j]; ++t) l_rows[(size_t)j * k + t] = l_rows[(size_t)j * k + t] * s->G + j;
            }
            ts = trag_merge_topk(l_rows.data(), l_scores.data(), l_n.data(), s->G, k, k, o_rows, o_scores, &out_n[b0 + b]);
            if (ts != TRAG_OK) return ts;
        }
    }
    return TRAG_OK;
}

trag_status trag_sharded_search(trag_sharded_store* s, const float* query, uint32_t dim, uint32_t k,
                                uint32_t* out_rows, float* out_scores, uint32_t* out_n) {
    return trag_sharded_search_batch(s, query, 1, dim, k, out_rows, out_scores, out_n);
}

trag_status trag_shard_rccl_version(int* version) {
    if (!version) return set_error(TRAG_INVALID_ARG, "null argument");
    const Rccl* R = nullptr;
    trag_status ts = need_rccl(&R);
    if (ts != TRAG_OK) return ts;
    NCCL_TRY(R, R->GetVersion(version));
    return TRAG_OK;
}

}  // extern "C"
