/*
 * trag.h -- C ABI of the MI355X (gfx950) dense-retrieval library.
 *
 * The reference (trueno-rag, Rust) has no FFI: the seam this library replaces
 * is the method surface of the concrete `VectorStore` (src/index.rs:323-437)
 * plus `FusionStrategy::fuse` (src/fusion.rs:42-63). Each entry point cites the
 * reference item a Rust binding would forward to it. Paths are relative to the
 * reference tree.
 *
 * Conventions
 *  - Plain pointers and sizes only. Unless a name ends in `_device`, every
 *    pointer is host memory owned by the caller; the library owns all device
 *    memory behind the opaque handle.
 *  - Rows are addressed by a u32 row index assigned at insert (insertion
 *    order). The binding keeps the ChunkId <-> row tables (ChunkId is a random
 *    UUID, src/chunk.rs:14-16, never compared numerically).
 *  - Scores are "higher is better" for every metric (Euclidean is negated,
 *    src/index.rs:400). Results come best first; equal scores are ordered by
 *    ascending row index; NaN scores come last.
 *  - Return value 0 = OK; the message of the last failure on the calling
 *    thread is available from trag_last_error().
 *  - Thread contract mirrors &self / &mut self: searches on one handle may be
 *    issued concurrently; insert/overwrite/remove/reserve need exclusive access.
 *    A handle's primary search scratch holds one search at a time: the
 *    host-pointer entries hold an internal lock until their results are back --
 *    but a host-pointer search of ONE query that finds the lock taken runs
 *    beside its holder in one of three spare contexts (own stream, scratch and
 *    staging; the plain route only: exact-order scan, nothing pending), so four
 *    threads on one handle overlap on the GPU (10 000 x 384: 3.2 x the queries
 *    per second of one thread); the `_device` entries hold the lock while they
 *    enqueue, and a call that
 *    arrives on another stream than the handle's previous call (a caller's
 *    stream, then the handle's own, or two caller streams) first makes the new
 *    stream wait for the work the previous call enqueued. A stream passed to a
 *    `_device` entry must stay alive until the handle's next call.
 *    (A handle in pipelined mode, trag_store_set_pipelined, holds two scratch sets and overlaps the
 *    select of one single-query search with the scan of the next; see there.)
 */
#ifndef TRAG_H
#define TRAG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct trag_store trag_store;

typedef enum trag_status {
    TRAG_OK = 0,
    TRAG_DIM_MISMATCH = 1,   /* Error::DimensionMismatch{expected,actual}, src/index.rs:365-370,387-392 */
    TRAG_INVALID_CONFIG = 2, /* Error::InvalidConfig, src/index.rs:360-363 */
    TRAG_DEVICE_ERROR = 3,   /* HIP failure -> Error::VectorStore(String), src/error.rs:38-39 */
    TRAG_OOM = 4,
    TRAG_INVALID_ARG = 5,
    TRAG_UNSUPPORTED = 6
} trag_status;

/* DistanceMetric, src/index.rs:310-319 */
typedef enum trag_metric { TRAG_COSINE = 0, TRAG_EUCLIDEAN = 1, TRAG_DOT_PRODUCT = 2 } trag_metric;

/* How the corpus scan is computed. Results are identical in every mode. */
typedef enum trag_scan_mode {
    TRAG_SCAN_AUTO = 0,
    TRAG_SCAN_EXACT = 1,  /* every row scored with the reference's sequential f32 order */
    TRAG_SCAN_STREAM = 2, /* reassociated streaming scan + exact re-score of candidates */
    TRAG_SCAN_MFMA = 3,   /* query batches as Q x D^T on the f32 matrix cores + exact re-score
                             (dim % 32 == 0); AUTO picks it above 8 queries where the bf16 pass
                             below does not apply; 2..8 queries on a short corpus share one
                             exact corpus pass */
    TRAG_SCAN_BF16 = 4,   /* query batches: a bf16 matrix-core pass picks, per query, every row
                             that can be in the top k (rigorous error bound), no score matrix is
                             written, and those rows are scored exactly in f32 (dim % 64 == 0,
                             at least 65 536 rows, k <= 256). AUTO uses it above 8 queries, and
                             from 2 queries up where the extra queries of a multi-query exact pass
                             would cost more than this pass's small launches (2 queries from about
                             1.4 M x 384 rows, 4 from 0.5 M, 8 from 0.2 M) */
    TRAG_SCAN_MIRROR = 5  /* needs the bf16 mirror (trag_store_set_bf16_mirror): single queries
                             scan the mirror (half the bytes), the rows that can be in the top
                             k under a rigorous bound are then scored exactly in f32; batches
                             run as TRAG_SCAN_BF16 reading the mirror. AUTO does the same once
                             the mirror exists. Without a mirror: as AUTO. */
} trag_scan_mode;

/* VectorStoreConfig, src/index.rs:283-307 (the hnsw_* fields are never read by
 * the reference and have no counterpart here). */
typedef struct trag_config {
    uint32_t dim;          /* VectorStoreConfig::dimension, default 384 */
    uint32_t metric;       /* trag_metric, default cosine */
    int32_t device;        /* HIP device ordinal; -1 = current device */
    uint32_t scan_mode;    /* trag_scan_mode */
    uint64_t reserve_rows; /* capacity hint, 0 = grow on demand */
} trag_config;

/* VectorStore::new / with_dimension, src/index.rs:335-350 */
trag_status trag_store_create(const trag_config *cfg, trag_store **out);
void trag_store_destroy(trag_store *s);
/* #[derive(Clone)] of VectorStore, src/index.rs:322: a second, independent store on the same
 * device with the same rows (removed rows stay removed, row indices are kept), configuration and
 * bf16 mirror setting. A device-to-device copy; scratch and learned plans are not copied. */
trag_status trag_store_clone(trag_store *src, trag_store **out);

/* VectorStore::config, src/index.rs:354-356 */
trag_status trag_store_config(const trag_store *s, trag_config *out);

/* VectorStore::insert / insert_batch, src/index.rs:359-383. Appends n rows of
 * `dim` floats (row-major). dim != configured dimension -> TRAG_DIM_MISMATCH
 * and nothing is inserted. rows == NULL with n > 0 -> TRAG_INVALID_CONFIG
 * ("chunk must have embedding"). *out_first_row receives the index of the
 * first appended row. A duplicate ChunkId is the binding's job: it calls
 * trag_store_overwrite on the existing row (the reference overwrites,
 * src/index.rs:372). */
trag_status trag_store_insert(trag_store *s, const float *rows, uint64_t n, uint32_t dim,
                              uint32_t *out_first_row);
trag_status trag_store_overwrite(trag_store *s, uint32_t row, const float *vec, uint32_t dim);

/* VectorStore::remove, src/index.rs:421-424. Tombstones the row: it is never
 * returned again; its index is not reused. Removing a dead or out-of-range
 * row is TRAG_INVALID_ARG (the binding maps that to None). */
trag_status trag_store_remove(trag_store *s, uint32_t row);

/* VectorStore::len / is_empty, src/index.rs:428-436: live rows. */
uint64_t trag_store_len(const trag_store *s);
/* Rows ever inserted (live + tombstoned) = next row index. */
uint64_t trag_store_rows(const trag_store *s);
trag_status trag_store_reserve(trag_store *s, uint64_t rows);

/* VectorStore::search, src/index.rs:386-412. dim != configured dimension ->
 * TRAG_DIM_MISMATCH. Writes *out_n = min(k, live rows) results; k == 0, an
 * empty store and k > len are not errors (src/index.rs:409). out_rows and
 * out_scores must hold k entries. */
trag_status trag_store_search(trag_store *s, const float *query, uint32_t dim, uint32_t k,
                              uint32_t *out_rows, float *out_scores, uint32_t *out_n);

/* The same for B queries (row-major B x dim), the shape HybridRetriever /
 * DenseRetriever callers produce when many queries are in flight
 * (src/retrieve.rs:181,225,286). Outputs are B x k with out_n[b] valid entries
 * in row b. */
trag_status trag_store_search_batch(trag_store *s, const float *queries, uint32_t n_queries,
                                    uint32_t dim, uint32_t k, uint32_t *out_rows,
                                    float *out_scores, uint32_t *out_n);

/* Device-resident variant: d_queries, d_out_* are device pointers on the
 * store's device; work is enqueued on `hip_stream` (a hipStream_t, NULL = the
 * default stream) and the call returns without waiting. d_status[b] (optional,
 * may be NULL) receives 0 when query b's result is final; a non-zero value
 * (1 ranking not certified, 2 candidate overflow, 3 crowded score bin needs the
 * multi-level select) means the fast path gave up and the caller must repeat
 * that query with trag_store_search / _batch, which handle all three
 * internally and remember the plan that worked for later calls. The handle also
 * learns from device-entry calls alone: the select kernel bumps a word in pinned
 * host memory whenever a query reports 3, the next call reads it (no
 * synchronisation), and once at least 4 queries and half of those enqueued so
 * far reported 3 the multi-level select becomes the plan. A caller that enqueues
 * hundreds of queries ahead of the GPU sees this take effect that much later.
 * Euclidean stores of fewer than about 1.8 M rows start on the multi-level select
 * (distances between normalised vectors crowd one bin as a rule; the form costs a
 * few microseconds there), so they do not go through this learning at all. */
trag_status trag_store_search_device(trag_store *s, const float *d_queries, uint32_t n_queries,
                                     uint32_t dim, uint32_t k, uint32_t *d_out_rows,
                                     float *d_out_scores, uint32_t *d_out_n,
                                     uint32_t *d_status, void *hip_stream);

/* Pipelined single-query searches on the device entry. VectorStore::search (src/index.rs:386-412) is
 * "score every row, then order": on the GPU a scan of the corpus and a one-workgroup select behind
 * it, which runs alone on the device (17 us behind a 235 us scan of 1 M x 384). A pipelined handle
 * keeps two search scratch sets and runs the select of a query beside the NEXT query's scan:
 *   enabled = 1  on the handle's own side stream (events order scan -> select -> the set's next use);
 *                a query's results exist soon after its own scan -- what an exchange that runs beside
 *                the next scan needs (trag_shard_config.overlap switches this on);
 *   enabled = 2  as one more workgroup of the next scan's launch: one launch per query, no event, no
 *                second stream (1 M x 384: 0.239 ms a query against 0.253 unpipelined and 0.248 with
 *                mode 1). The select of the LAST query enqueued stays pending inside the handle until
 *                the next query's launch carries it or trag_store_fence (or any call that takes
 *                another route, changes the stream or replaces the scratch) launches it on its own.
 *   enabled = 0  off (what is pending is launched, what runs on the side stream is waited for).
 * Only trag_store_search_device calls with one query that take the exact-order scan and the
 * one-workgroup select are pipelined; every other route first settles the pipeline and then runs
 * on the caller's stream as before. The outputs of a pipelined call (rows, scores, count, status) may
 * be read on a stream only after trag_store_fence on it, and the query and the output buffers must stay
 * valid until then. On a store with the bf16 mirror, single queries take the scan of the mirror and the three
 * small stages behind it (prune, exact scores, ordering) go to the side stream in either mode (1 M x 384:
 * 0.136 ms a query against 0.154); so do the stages of the multi-level select where that is the handle's plan. Host-pointer entries are unaffected (they wait for their own results). Costs a second scratch
 * set (8 bytes per row instead of 4). No counterpart in the reference. */
trag_status trag_store_set_pipelined(trag_store *s, int enabled);
/* Makes hip_stream wait for every select the handle still owes: a pending one (mode 2) is launched
 * on the stream of its scan, those on the side stream (mode 1) are waited for. A no-op on a handle
 * that is not pipelined. Enqueues work and waits, does not block the host. */
trag_status trag_store_fence(trag_store *s, void *hip_stream);

/* No counterpart in the reference. For callers that cannot allocate device memory -- code
 * under HIP graph capture, serving loops with a frozen memory plan, harnesses that forbid
 * hipMalloc: trag_store_search_device for single-query exact scans with the search scratch
 * (per-row keys, tile and wave maxima, histograms, candidate lists) in a fixed 65.3 MB device
 * array inside the library instead of per-handle allocations. Same arguments, results,
 * d_status codes and ordering rule; queries are scanned one at a time in the reference's
 * summation order (no matrix-core or mirror route). It calls no hipMalloc / hipFree / event or
 * stream creation. Limits: at most 16 777 216 rows ever inserted in the store and k <= 4064
 * (unless the store holds <= 4096 live rows), else TRAG_UNSUPPORTED. All calls on one GPU share
 * that GPU's array, across stores: calls whose GPU work overlaps (two streams, two threads with
 * two stores) race and return garbage -- keep them on one stream or order them with events.
 * Under stream capture (hipStreamBeginCapture on hip_stream) the call records its launches and
 * memsets into the graph and touches no other stream; a replay with other queries in the same
 * device buffers returns what the call itself would. Rows inserted since the handle's last search
 * make a captured call TRAG_UNSUPPORTED (run one search outside the capture first), and the
 * handle does not order its later calls behind replays: that is the caller's. */
trag_status trag_store_search_device_noalloc(trag_store *s, const float *d_queries,
                                             uint32_t n_queries, uint32_t dim, uint32_t k,
                                             uint32_t *d_out_rows, float *d_out_scores,
                                             uint32_t *d_out_n, uint32_t *d_status, void *hip_stream);

/* Append n rows produced on the device by the library's counter-based
 * synthetic generator (seeded, unit-norm rows); used by benchmarks so that a
 * multi-GB corpus never crosses PCIe. Row i of the call is row
 * first_global_row + i of the generator's stream, so the shards of a
 * row-partitioned corpus are slices of one global corpus. No counterpart in
 * the reference. */
trag_status trag_store_fill_synthetic(trag_store *s, uint64_t n, uint64_t seed,
                                      uint64_t first_global_row, uint32_t *out_first_row);
/* Copy rows [row0, row0+n) back to host memory (n x dim floats). */
trag_status trag_store_read_rows(const trag_store *s, uint64_t row0, uint64_t n, float *out);

/* FusionStrategy::fuse, src/fusion.rs:42-63. strategy: 0 RRF{k=param}
 * (:68-84), 1 Linear{dense_weight=param} (:87-109), 2 Convex{alpha=param}
 * (:112-119), 3 DBSF (:122-138), 4 Union (:141-160), 5 Intersection
 * (:163-180). Ids are opaque u64 tokens (the binding passes row indices or its
 * own id table positions). Writes *out_n <= nd + ns fused entries, best first
 * (src/fusion.rs:227-231); equal fused scores keep first-appearance order
 * (dense list, then sparse list). Pure host code: the inputs are <= 100
 * entries (src/retrieve.rs:94,176). */
trag_status trag_fuse(uint32_t strategy, float param, const uint64_t *dense_ids,
                      const float *dense_scores, uint32_t nd, const uint64_t *sparse_ids,
                      const float *sparse_scores, uint32_t ns, uint64_t *out_ids,
                      float *out_scores, uint32_t *out_n);
/* Shorthand for the default strategy, RRF (src/fusion.rs:33-37): scores of the
 * inputs are ignored, only their order matters. */
trag_status trag_fuse_rrf(const uint64_t *dense_ids, uint32_t nd, const uint64_t *sparse_ids,
                          uint32_t ns, float k, uint64_t *out_ids, float *out_scores,
                          uint32_t *out_n);

/* RRF (src/fusion.rs:68-84) for a batch of queries whose dense and sparse result lists are in
 * device memory (trag_store_search_device, trag_bm25_search_device): the hybrid step of
 * src/retrieve.rs:175-220 without a host round trip, n_queries fusions in one launch. Ids are u32 in
 * one id space (row index = document index: every chunk went into both indices in the same
 * order). Query b's dense list is d_dense_ids[b * dense_stride .. + d_dense_n[b]) (d_dense_n NULL:
 * dense_stride entries), its sparse list likewise; at most 1024 entries per list. Writes, per
 * query, the best min(k_out, distinct ids) entries at b * k_out and their count. Same arithmetic
 * and order as trag_fuse. Enqueued on hip_stream, no synchronisation. */
trag_status trag_fuse_rrf_device(const uint32_t *d_dense_ids, const uint32_t *d_dense_n,
                                 uint32_t dense_stride, const uint32_t *d_sparse_ids,
                                 const uint32_t *d_sparse_n, uint32_t sparse_stride,
                                 uint32_t n_queries, float k, uint32_t k_out, uint32_t *d_out_ids,
                                 float *d_out_scores, uint32_t *d_out_n, void *hip_stream);

/* Merge per-shard top-k lists (row-sharded corpus, one list per shard, global
 * row ids) into the global top-k with the same ordering rule as search.
 * lists are n_lists x k_in, list_n[i] valid entries each. Host code. */
trag_status trag_merge_topk(const uint32_t *rows, const float *scores, const uint32_t *list_n,
                            uint32_t n_lists, uint32_t k_in, uint32_t k_out, uint32_t *out_rows,
                            float *out_scores, uint32_t *out_n);

/* Device-side variant of trag_merge_topk for lists that an all-gather left in
 * device memory: d_rows / d_scores are n_lists x k_in, d_list_n (may be NULL =
 * all full) n_lists counts; n_lists * k_in <= 4096. Enqueued on hip_stream. */
trag_status trag_merge_topk_device(const uint32_t *d_rows, const float *d_scores,
                                   const uint32_t *d_list_n, uint32_t n_lists, uint32_t k_in,
                                   uint32_t k_out, uint32_t *d_out_rows, float *d_out_scores,
                                   uint32_t *d_out_n, void *hip_stream);

/* Wire format of the cross-shard exchange, per query: k score bit patterns,
 * k GLOBAL row ids, 1 count = 2k+1 32-bit words. trag_pack_topk_device writes
 * n_queries such records from a search result (adding first_global_row to the
 * local row ids); after an all-gather the receive buffer holds
 * [n_lists][n_queries][2k+1] words and trag_merge_packed_device reduces it to
 * the global top-k per query (same ordering rule as search). n_lists * k <= 4096. Bits 16..23 of
 * a record's count word may carry the status of that query on that shard (0 = final); the merge
 * ignores them here (trag_shard_search_device reports them). */
trag_status trag_pack_topk_device(const uint32_t *d_rows, const float *d_scores, const uint32_t *d_n,
                                  uint32_t first_global_row, uint32_t n_queries, uint32_t k,
                                  uint32_t *d_packed, void *hip_stream);
trag_status trag_merge_packed_device(const uint32_t *d_gathered, uint32_t n_lists,
                                     uint32_t n_queries, uint32_t k, uint32_t *d_out_rows,
                                     float *d_out_scores, uint32_t *d_out_n, void *hip_stream);

/* ------------------------------------------------------------------------
 * Sharded search: the corpus partitioned by rows over the GPUs of one node.
 * No counterpart in the reference (single process, one thread); the call this
 * serves is the dense search of HybridRetriever::retrieve (src/retrieve.rs:181)
 * and of DenseRetriever::retrieve (:286) when one GPU's HBM or bandwidth is
 * not enough. Every shard answers the query batch from its own rows, ONE
 * all-gather (RCCL over xGMI) moves the per-shard lists in the wire format
 * above, and every GPU merges them with the ordering rule of a single store.
 * RCCL is looked up at run time (dlopen of librccl.so.1; TRAG_RCCL_LIB names
 * another file): the process decides which copy it runs.
 * ------------------------------------------------------------------------ */

/* (a) One rank per process (one GPU each), the launch model of
 * torch.distributed / MPI. Rank 0 makes an id, the host program carries its
 * TRAG_SHARD_ID_BYTES bytes to the other ranks, every rank creates its group
 * (a collective call: ncclCommInitRank). */
#define TRAG_SHARD_ID_BYTES 128
typedef struct trag_shard_group trag_shard_group;
typedef struct trag_shard_config {
    uint32_t rank, world;
    int32_t device;       /* HIP device ordinal of this rank; -1 = current device */
    uint32_t k_max;       /* largest k of later searches; world * k_max <= 4096 */
    uint32_t max_queries; /* largest batch of one trag_shard_search_device call */
    uint32_t overlap;     /* 0: pack, all-gather and merge follow the local search on the caller's
                             stream. 1: they run on the group's own stream, so the next call's scan
                             overlaps this call's exchange -- and the local store is switched to
                             deferred selects (trag_store_set_pipelined, mode 2): the select of a
                             single-query call rides in the next call's scan launch, and that next
                             call (or trag_shard_fence, or any other call on the group) enqueues
                             the exchange of the one before. The caller's stream then carries one
                             launch and one event record per call (1.25 M x 384 per rank: 0.299 ms a
                             call with the exchange against 0.293 for the scan kernel alone).
                             Results are visible to a stream only after trag_shard_fence on it;
                             output buffers must stay valid until then. The store keeps that mode
                             when the group is done with it: searches made on it directly afterwards
                             need trag_store_fence, or trag_store_set_pipelined(store, 0) first. */
} trag_shard_config;
trag_status trag_shard_unique_id(uint8_t *id /* TRAG_SHARD_ID_BYTES */);
trag_status trag_shard_group_create(const trag_shard_config *cfg, const uint8_t *id, trag_shard_group **out);
void trag_shard_group_destroy(trag_shard_group *g);

/* Collective: every rank calls it with the same queries, n_queries and k. `local` is this rank's
 * shard; its local row r is global row row0 + r * row_mul (contiguous shards: row0 = first global
 * row, row_mul = 1; shards striped over the ranks: row0 = rank, row_mul = world). One call
 * enqueues local search -> records -> all-gather -> merge; every rank receives the global top k
 * (global row ids) in its own d_out_* arrays. d_status[b] (optional) is the worst status any
 * shard reported for query b (codes of trag_store_search_device): non-zero asks for
 * trag_shard_search_batch. Device pointers, no synchronisation. A rank whose local search fails
 * still joins the all-gather (with empty records that carry status 3) before it returns its error,
 * so its peers are not left waiting in the collective. */
trag_status trag_shard_search_device(trag_shard_group *g, trag_store *local, uint32_t row0, uint32_t row_mul,
                                     const float *d_queries, uint32_t n_queries, uint32_t dim, uint32_t k,
                                     uint32_t *d_out_rows, float *d_out_scores, uint32_t *d_out_n,
                                     uint32_t *d_status, void *hip_stream);
/* Makes hip_stream wait for every exchange enqueued so far (needed with overlap = 1). */
trag_status trag_shard_fence(trag_shard_group *g, void *hip_stream);
/* Host pointers, synchronous, any n_queries; finishes every query (the shard's host entry
 * repeats what its fast path gave up). Collective as well. */
trag_status trag_shard_search_batch(trag_shard_group *g, trag_store *local, uint32_t row0, uint32_t row_mul,
                                    const float *queries, uint32_t n_queries, uint32_t dim, uint32_t k,
                                    uint32_t *out_rows, float *out_scores, uint32_t *out_n);
/* NCCL-style version code of the RCCL in use (e.g. 22707). */
trag_status trag_shard_rccl_version(int *version);
/* What the group's RCCL communicator reports about itself (ncclCommCount, ncclCommUserRank,
 * ncclCommCuDevice): the number of ranks it spans, this rank's number in it and the device it is bound
 * to -- evidence, for a benchmark record, that the exchange really ran over `world` ranks. Any
 * output pointer may be NULL. TRAG_UNSUPPORTED when the RCCL in use lacks those calls. */
trag_status trag_shard_group_info(trag_shard_group *g, uint32_t *comm_world, uint32_t *comm_rank,
                                  int32_t *comm_device);

/* (b) One process, n GPUs: a VectorStore striped over device_ids[0..n_devices) -- global row g
 * lives on shard g % n as local row g / n, so appends keep the shards level, global row ids
 * never move and ascending global row stays insertion order (the tie rule carries over). One
 * communicator over all devices (ncclCommInitAll), one stream per GPU. cfg->device is ignored,
 * cfg->reserve_rows is the whole store's. The method surface is VectorStore's
 * (src/index.rs:335-437), as trag_store_*. */
typedef struct trag_sharded_store trag_sharded_store;
typedef enum trag_exchange {
    TRAG_EXCHANGE_RCCL = 0, /* queries by ncclBroadcast, records by ncclAllGather */
    TRAG_EXCHANGE_COPY = 1  /* the same exchange as device-to-device copies ordered by events: for
                               boxes without RCCL and for device lists RCCL refuses (the same GPU
                               named twice, which is how the n > 1 logic is tested on one GPU) */
} trag_exchange;
trag_status trag_sharded_create(const trag_config *cfg, uint32_t n_devices, const int32_t *device_ids,
                                uint32_t exchange, trag_sharded_store **out);
void trag_sharded_destroy(trag_sharded_store *s);
trag_status trag_sharded_insert(trag_sharded_store *s, const float *rows, uint64_t n, uint32_t dim,
                                uint32_t *out_first_row);
trag_status trag_sharded_overwrite(trag_sharded_store *s, uint32_t row, const float *vec, uint32_t dim);
trag_status trag_sharded_remove(trag_sharded_store *s, uint32_t row);
uint64_t trag_sharded_len(const trag_sharded_store *s);
uint64_t trag_sharded_rows(const trag_sharded_store *s);
trag_status trag_sharded_search(trag_sharded_store *s, const float *query, uint32_t dim, uint32_t k,
                                uint32_t *out_rows, float *out_scores, uint32_t *out_n);
trag_status trag_sharded_search_batch(trag_sharded_store *s, const float *queries, uint32_t n_queries,
                                      uint32_t dim, uint32_t k, uint32_t *out_rows, float *out_scores,
                                      uint32_t *out_n);
/* Global rows [first, first + n) of the library's synthetic generator (trag_store_fill_synthetic). */
trag_status trag_sharded_fill_synthetic(trag_sharded_store *s, uint64_t n, uint64_t seed, uint32_t *out_first_row);
uint32_t trag_sharded_devices(const trag_sharded_store *s);
/* Shard i's store, for per-shard settings (scan mode, bf16 mirror, statistics). Not for insert. */
trag_store *trag_sharded_shard(trag_sharded_store *s, uint32_t i);

/* Optional bf16 copy of the corpus in HBM (1.5 x the memory, dimension % 8 == 0),
 * kept in step with insert / overwrite / fill. It halves the bytes a prefiltering
 * pass reads; results stay the exact f32 results (see TRAG_SCAN_MIRROR). enabled
 * != 0 builds it from the rows present, 0 frees it. Needs exclusive access like
 * insert. No counterpart in the reference. */
trag_status trag_store_set_bf16_mirror(trag_store *s, int enabled);
int trag_store_has_bf16_mirror(const trag_store *s);

/* Change the scan mode of later searches (trag_scan_mode). */
trag_status trag_store_set_scan_mode(trag_store *s, uint32_t scan_mode);

/* Scan-kernel timing with HIP events on the stream the kernel runs on. While
 * enabled, every enabled-th scan launch (1 = every launch) is bracketed by an
 * event pair (at most 4096 are kept; an event pair costs about 6 us of stream
 * time, which matters for sub-millisecond scans). On a handle with deferred selects
 * (trag_store_set_pipelined mode 2) a step is one launch and the launches follow one another without
 * a gap, which a pair around a single launch would open; there a pair spans a run of
 * max(enabled, 4) consecutive launches and counts as that many. trag_store_scan_time waits for the
 * recorded events, returns their summed duration and the number of launches they span, and clears them. */
trag_status trag_store_set_profiling(trag_store *s, int enabled);
trag_status trag_store_scan_time(trag_store *s, double *total_ms, uint32_t *launches);

/* Measurement aid: streams the store's rows once per repeat with plain contiguous
 * 16-byte loads (no arithmetic) and reports the best rate in GB/s -- the memory
 * system's streaming-read rate on this device, beside which the scan's rate can be
 * read. No counterpart in the reference. */
trag_status trag_store_stream_probe(trag_store *s, uint32_t repeats, double *gb_per_s);

/* Counters of the last search on this handle (diagnostics). */
typedef struct trag_search_stats {
    uint32_t mode_used;      /* trag_scan_mode actually run */
    uint32_t candidates;     /* rows that reached the final sort */
    uint32_t refine_levels;  /* radix-select levels needed (1..3) */
    uint32_t fallbacks;      /* queries repeated with the exact scan */
    uint32_t keys_kept_from; /* ordered key below which the scan skipped its per-row key stores (long
                                single-query scans keep only what the select can read); 0: none skipped */
} trag_search_stats;
trag_status trag_store_last_stats(const trag_store *s, trag_search_stats *out);

/* ------------------------------------------------------------------------
 * BM25 sparse index (the sparse half of the hybrid retrieve,
 * src/retrieve.rs:183-187). Replaces `BM25Index` (src/index.rs:31-51) behind
 * its `SparseIndex` trait (src/index.rs:8-28).
 *
 * The boundary works on TERM IDS: the host side keeps the reference's
 * tokenizer (`BM25Index::tokenize`, src/index.rs:111-124) and a term -> u32
 * dictionary, and hands documents and queries over as id sequences.
 * Documents are numbered in insertion order (u32), like store rows. Postings
 * (CSR by term: document, term frequency) and document lengths live in HBM;
 * a query is scored there and the top k selected with the dense path's select.
 * ------------------------------------------------------------------------ */
typedef struct trag_bm25 trag_bm25;

typedef struct trag_bm25_config {
    float k1;        /* term-frequency saturation; reference default 1.2 (src/index.rs:69) */
    float b;         /* length normalisation; reference default 0.75 (src/index.rs:70) */
    int32_t device;  /* HIP device ordinal; -1 = current device */
} trag_bm25_config;

/* `BM25Index::new` / `with_params` (src/index.rs:62-84). */
trag_status trag_bm25_create(const trag_bm25_config *cfg, trag_bm25 **out);
void trag_bm25_destroy(trag_bm25 *ix);

/* `SparseIndex::add` / `add_batch` (src/index.rs:176-210) for n_docs tokenised
 * documents: document i owns terms[doc_offsets[i] .. doc_offsets[i+1]) (its
 * tokens in order, repeats included; its length is that token count). Returns
 * the index of the first new document. An empty document still counts
 * towards the document count and the average length, as in the reference. */
trag_status trag_bm25_add(trag_bm25 *ix, const uint32_t *terms, const uint64_t *doc_offsets,
                          uint64_t n_docs, uint32_t *out_first_doc);

/* `SparseIndex::remove` (src/index.rs:245-275): the document stops matching
 * and no longer counts in document frequencies, the document count or the
 * average length. Removing an unknown or already removed document is not an
 * error. */
trag_status trag_bm25_remove(trag_bm25 *ix, uint32_t doc);

/* `SparseIndex::len` (src/index.rs:277-279): documents added and not removed. */
uint64_t trag_bm25_len(const trag_bm25 *ix);

/* `SparseIndex::search` (src/index.rs:212-243) for a tokenised query (its
 * terms in order, repeats included: a repeated term counts twice, as in the
 * reference). Writes min(k, documents with score > 0) results, best first;
 * equal scores by ascending document index. Unknown terms match nothing; an
 * empty query returns nothing. */
trag_status trag_bm25_search(trag_bm25 *ix, const uint32_t *query_terms, uint32_t n_terms, uint32_t k,
                             uint32_t *out_docs, float *out_scores, uint32_t *out_n);

/* Same search with device-resident outputs (k entries + 1 count + 1 status
 * word), enqueued on hip_stream without synchronising. *d_status follows
 * trag_store_search_device's codes; 3 asks the caller to use trag_bm25_search
 * for this query (crowded score bins need the multi-level select). */
trag_status trag_bm25_search_device(trag_bm25 *ix, const uint32_t *query_terms, uint32_t n_terms,
                                    uint32_t k, uint32_t *d_out_docs, float *d_out_scores,
                                    uint32_t *d_out_n, uint32_t *d_status, void *hip_stream);

/* Benchmark input: n_docs synthetic documents of tokens_per_doc tokens drawn
 * from Zipf(1.0) over vocab terms (term 0 the most frequent), token p of
 * document i a pure function of (seed, first_global_doc + i, p). No
 * counterpart in the reference. */
trag_status trag_bm25_fill_synthetic(trag_bm25 *ix, uint64_t n_docs, uint32_t vocab,
                                     uint32_t tokens_per_doc, uint64_t seed,
                                     uint64_t first_global_doc, uint32_t *out_first_doc);

/* Diagnostics: postings held, distinct terms with at least one posting. */
trag_status trag_bm25_stats(trag_bm25 *ix, uint64_t *postings, uint64_t *terms);

/* `MockEmbedder::hash_to_vector` (src/embed.rs:124-155), the input generator
 * of the reference's tests and of benches/retrieval.rs: dim values from one
 * running SipHash-1-3 (std's DefaultHasher, zero keys) over (text, i) pairs,
 * mapped to [-1, 1], L2-normalised when `normalize` is nonzero
 * (EmbeddingConfig::normalize, default true, src/embed.rs:41-48). Host code;
 * needs no GPU. The reference pins only length, unit norm, determinism and
 * text sensitivity of these vectors (src/embed.rs:992-1036), not their values. */
trag_status trag_mock_embed(const char *text, uint64_t len, uint32_t dim, int normalize, float *out);

const char *trag_last_error(void);
const char *trag_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TRAG_H */
