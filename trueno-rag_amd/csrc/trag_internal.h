// Internal declarations shared by the host API and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include "trag.h"
#include "trag_error.h"

namespace trag {

// Radix-select geometry: three levels over the 32-bit ordered key, 11 + 11 + 10 bits.
constexpr uint32_t HIST_BINS = 2048;
constexpr uint32_t HIST_LEVELS = 3;
// Most rows that may reach the final single-workgroup sort.
constexpr uint32_t CAND_CAP = 4096;
// Extra candidates beyond k requested by the streaming mode, so that the
// exactly re-scored top-k can be certified against everything left behind.
constexpr uint32_t STREAM_SLACK = 32;
// Most waves a scan grid may have (sizes the per-wave maxima).
constexpr uint32_t WMAX_CAP = 16384;
// Scan waves that can take part in the sparse key stores (see trag_scan_common.h): one entry each.
constexpr uint32_t SPARSE_WAVES = 1024;
// ... and the scans long enough for them to pay: the polls cost a wave ~2 us each and the first
// tiles of every wave are stored regardless. MI355X, scan time with / without: exact 10M x 384
// 2.262 / 2.382 ms, mirror 10M x 384 1.170 / 1.262; switched on for every size: 1M x 384 exact
// 0.249 / 0.238, 1.25M x 384 0.303 / 0.295, 2M x 768 0.912 / 0.917, 2M x 4096 unchanged (with
// one polling wave per workgroup instead of all four: 0.245 / 0.240, 0.299 / 0.295, 0.909 / 0.918).
// Exact scan at 384 dimensions by rows, with / without: 2M 0.460 / 0.466, 2.5M 0.575 / 0.582, 3.3M 0.745 / 0.766:
// on from 28 tiles per wave.
inline bool sparse_scan_pays(uint32_t n_tiles, uint32_t n_waves) { return n_waves > 0 && n_tiles / n_waves >= 28u; }

// Final status written per query.
constexpr uint32_t ST_OK = 0;
constexpr uint32_t ST_UNCERTIFIED = 1;  // streaming scan: margin check failed, repeat exactly
constexpr uint32_t ST_NEED_REFINE = 3;  // one-workgroup select: coarse bin too crowded, use the full select

// Per-query control block in device memory. Written by prep_kernel and by
// block 0 of each select stage; read by the stages after it on the same stream.
struct SearchCtl {
    float na;            // ||q|| with the reference's sequential f32 sum and sqrt
    float na_sq;         // sum q*q, same order
    float vmax_sq;       // upper bound of max_row sum v*v (streaming error bound)
    uint32_t need;       // rows wanted at the final sort: min(k + slack, live)
    uint32_t k;          // results wanted: min(k, live)
    uint32_t resolved;   // 0 while the threshold is open, else the level that fixed it
    uint32_t b1, above1; // level-1 bin holding the need-th key, rows strictly above it
    uint32_t b2, above2; // level 2, cumulative
    uint32_t t_key;      // accept keys >= t_key (or > t_key when cut_ties)
    uint32_t cut_ties;   // 1: more than CAND_CAP rows tie on the need-th key
    uint32_t tie_quota;  // ties to take, lowest rows first
    uint32_t above3;     // rows strictly above t_key when cut_ties
    uint32_t expect;     // candidates the final stage must find
    uint32_t cand_count; // atomic append cursor
    uint32_t status;
    uint32_t levels;     // radix levels the last select needed (diagnostics)
    uint32_t n_waves;    // waves of the scan grid: tile t was scanned by wave t % n_waves
    uint32_t keep_key;   // sparse key stores: the bound the select raised its cut to (0: none; diagnostics)
    float rho;           // bf16 prefilter: measured relative rounding error of the operands (trag_device.h: bf16_operand_rho)
};

// Device scratch of one in-flight query.
struct Scratch {
    // ctl and hist live in one allocation ("zero block"): they are what a
    // scan accumulates into and must be zero when a query starts.
    SearchCtl* ctl;      // 1
    uint32_t* hist;      // HIST_LEVELS * HIST_BINS
    uint32_t* wmax;      // WMAX_CAP: largest key each scan wave saw (not an accumulator)
    uint32_t* cand;      // CAND_CAP row indices
    uint32_t* ckeys;     // CAND_CAP exact ordered keys of the candidates (mirror scan, multi-level form)
    uint32_t* keys;      // capacity rows, ordered 32-bit keys (0 = removed row)
    uint32_t* tmax;      // capacity rows / 64: largest key of each 64-row tile
    uint32_t* events;    // host-pinned, device-visible word (may be null): queries whose one-workgroup
                         // select asked for the multi-level form; the host reads it without synchronising
    unsigned long long* wbest = nullptr;  // SPARSE_WAVES entries (scan epoch << 32) | best key a wave has published
    uint32_t epoch = 0;  // tag of the current scan in wbest[]; 0: the scan stores every row's key
};

// Device scratch of a block of queries on the batched (matrix-core) path.
struct BatchScratch {
    float* qpad;         // queries zero-padded to a multiple of 128 rows
    SearchCtl* ctls;     // one per query
    uint32_t* keys;      // [queries][keys_stride]
    size_t keys_stride;
    uint32_t* tmax;      // [queries][tmax_stride]
    size_t tmax_stride;
    uint32_t* cand;      // [queries][CAND_CAP]
};

// bf16 prefilter, thresholds that tighten during the full pass (trag_device.h: dyn_hit_level).
// Every query has a ladder of DYN_LEVELS thresholds, DYN_SUB steps per 2.02 eps, above the one its
// sample gave; a hit is counted for at most DYN_MAXINC levels. (1.25 M x 384 x 1000 queries, one
// device, ms per step: 4 steps per 2.02 eps 1.260, 2 steps 1.237; counted for 8 levels 1.275.)
#ifndef TRAG_DYN_LEVELS
#define TRAG_DYN_LEVELS 64
#define TRAG_DYN_SUB 2
#define TRAG_DYN_MAXINC 4
#endif
constexpr uint32_t DYN_LEVELS = TRAG_DYN_LEVELS;
constexpr uint32_t DYN_SUB = TRAG_DYN_SUB;
constexpr uint32_t DYN_MAXINC = TRAG_DYN_MAXINC;

// Device scratch of a block of queries on the bf16-prefilter path (no score matrix).
struct Bf16Scratch {
    uint16_t* qb;        // queries as bf16, zero-padded to a multiple of 256 rows
    float* thr;          // per query: the full pass's candidate threshold (the sample's; after the prune stage: the final one)
    uint32_t* dyn_lvl;   // per query (padded to 256): the ladder level the full pass has reached (only grows)
    float* dyn_lad;      // per query (padded): the ladder, two floats: level 0 and the step
    uint32_t* dyn_cnt;   // [queries][DYN_LEVELS]: rows seen that vouch for each level
    SearchCtl* ctls;     // one per query
    uint32_t* tmax;      // [queries][tmax_stride]: tile maxima of the sample pass
    size_t tmax_stride;
    uint32_t* cand;      // [queries][CAND_CAP]
    float* rowc;         // [rows rounded up to 64][2]: per row (scale, shift) of the query-stationary form's score test
    uint32_t* ckeys;     // [queries][CAND_CAP]: first the candidates' approximate values (prefilter -> prune), then
                         // their exact ordered keys (exact finish)
};

constexpr uint32_t ST_OVERFLOW = 2;     // more candidates than the final sort holds: repeat exactly
constexpr uint32_t BATCH_QUERIES = 256; // queries processed per batched pass

// Queries a multi-query exact scan handles in one corpus pass.
constexpr uint32_t MULTI_QUERIES = 8;
// LDS budget of a multi-query pass (of the CU's 160 KB) and its static part: 4 wave tiles of
// 64 x 36 floats, one coarse histogram and one norm per query; the rest holds the queries.
constexpr uint64_t MULTI_LDS_TOTAL = 158 * 1024;
constexpr uint64_t multi_static_lds(uint32_t nq) {
    return 4ull * 64 * 36 * 4 + (uint64_t)nq * HIST_BINS * 4 + (uint64_t)nq * 4 + 64;
}

// Distance between consecutive queries' scratch in a multi-query scan: query j's
// control block + histograms start j * zero_bytes BYTES behind the first one's,
// its keys / tile maxima / wave maxima j * keys / tmax / wmax ELEMENTS behind.
struct MultiStride {
    size_t zero_bytes = 0;
    size_t keys = 0;
    size_t tmax = 0;
    size_t wmax = 0;
};

struct ScanArgs {
    const float* corpus;
    const uint32_t* dead;  // tombstone bitmap, 1 bit per row
    uint32_t n_rows;
    uint32_t dim;
    uint32_t metric;
};


// trag_store_fill_synthetic with a step between the generator rows: row i of the call is row
// first_global_row + i * row_step of the generator's stream (shards striped over several GPUs)
trag_status store_fill_synthetic(trag_store* s, uint64_t n, uint64_t seed, uint64_t first_global_row,
                                 uint64_t row_step, uint32_t* out_first_row);

// What the one-workgroup select of a single exact-order query needs: kept by a handle whose selects are
// deferred (the select of query i runs as one extra workgroup of query i + 1's scan launch, or alone
// when something needs the results first).
struct PrevSelect {
    const float* q = nullptr;
    uint32_t n_rows = 0, d = 0, live = 0, need = 0, k = 0, n_waves = 0;
    SearchCtl* ctl = nullptr;
    uint32_t* hist = nullptr;
    const uint32_t* keys = nullptr;
    const uint32_t* tmax = nullptr;
    const uint32_t* wmax = nullptr;
    uint32_t* out_rows = nullptr;
    float* out_scores = nullptr;
    uint32_t* out_n = nullptr;
    uint32_t* out_status = nullptr;
    uint32_t* events = nullptr;
    const unsigned long long* wbest = nullptr;
    uint32_t epoch = 0;
};

// Whether the handle holds a deferred select that no launch carries yet (trag_store_set_pipelined(s, 2):
// the results of the last single query do not exist before the next scan's launch or a fence).
bool store_select_pending(trag_store* s);
// Makes `st` wait for the stages a pipelined handle has on its side stream (the mirror scan's, the multi-level
// select's) without launching a deferred select that is still pending (trag_store_fence does both).
trag_status store_wait_side(trag_store* s, hipStream_t st);

// ---- kernel launchers (trag_kernels.hip) ----
// the scans report the number of waves of their grid (tile t is scanned by wave t % n_waves)
// prev (scan_can_fuse(dim) only): that select rides in the scan's launch as one more workgroup
bool scan_can_fuse(uint32_t dim);
hipError_t launch_scan_exact(const float* d_q, const ScanArgs& a, const Scratch& s, uint32_t need,
                             uint32_t k, const float* d_vmax_sq, uint32_t* n_waves, hipStream_t st,
                             const PrevSelect* prev = nullptr);
hipError_t launch_scan_stream(const float* d_q, const ScanArgs& a, const Scratch& s, uint32_t need,
                              uint32_t k, const float* d_vmax_sq, uint32_t* n_waves, hipStream_t st);
// n_queries > 1: one workgroup per query of a multi-query scan (scratch strided by *ms, the
// queries d apart, outputs out_stride entries apart)
hipError_t launch_select_final(const float* d_q, const ScanArgs& a, const Scratch& s, bool rescore,
                               uint32_t live, uint32_t need, uint32_t k, uint32_t n_waves,
                               uint32_t* d_out_rows, float* d_out_scores, uint32_t* d_out_n,
                               uint32_t* d_status, hipStream_t st, uint32_t n_queries = 1,
                               const MultiStride* ms = nullptr, uint32_t out_stride = 0);
// exact scan of 2..MULTI_QUERIES queries (rows of d_q) in one corpus pass; s is query 0's scratch
uint32_t multi_group_limit(uint32_t dim);
hipError_t launch_scan_exact_multi(const float* d_q, uint32_t n_queries, const ScanArgs& a,
                                   const Scratch& s, const MultiStride& ms, uint32_t need, uint32_t k,
                                   const float* d_vmax_sq, uint32_t* n_waves, hipStream_t st);
bool stream_supports_dim(uint32_t dim);
hipError_t launch_select(const ScanArgs& a, const Scratch& s, hipStream_t st);
hipError_t launch_reset_ctl(const Scratch& s, hipStream_t st);
hipError_t launch_full_sort(const ScanArgs& a, const Scratch& s, unsigned long long* d_sort,
                            unsigned long long padded, uint32_t n_out, uint32_t* d_out_rows,
                            float* d_out_scores, uint32_t* d_out_n, uint32_t* d_status,
                            hipStream_t st);
hipError_t launch_final(const float* d_q, const ScanArgs& a, const Scratch& s, bool rescore,
                        uint32_t live, uint32_t* d_out_rows, float* d_out_scores, uint32_t* d_out_n,
                        uint32_t* d_status, hipStream_t st);
// batched path (trag_batch.hip + the batched forms of compact/final)
bool batch_supports(uint32_t dim, uint32_t metric);
hipError_t launch_batch_scan(const float* d_Q, uint32_t n_queries, const ScanArgs& a,
                             const float* d_norm2, const BatchScratch& s, uint32_t need, uint32_t k,
                             const float* d_vmax_sq, hipStream_t st);
hipError_t launch_batch_finish(const float* d_Q, uint32_t n_queries, const ScanArgs& a,
                               const BatchScratch& s, uint32_t live, uint32_t out_stride,
                               uint32_t* d_out_rows, float* d_out_scores, uint32_t* d_out_n,
                               uint32_t* d_status, hipStream_t st);
// pieces of the batched path reused by the bf16 prefilter
// one launch before the prefilter's first pass: norms and control blocks of the queries, the queries as
// bf16 (zero rows up to `padded`), and with_rowc the per-row (scale, shift) table
hipError_t launch_bf16_prep(const float* d_Q, uint32_t n_queries, uint32_t padded, const ScanArgs& a,
                            const float* d_norm2, const Bf16Scratch& s, uint32_t need, uint32_t k,
                            const float* d_vmax_sq, bool with_rowc, hipStream_t st);
// bf16 prefilter (trag_bf16.hip) and its exact finish (trag_kernels.hip)
bool bf16_supports(uint32_t dim, uint32_t metric);
uint32_t bf16_sample_step(uint64_t n_rows, uint32_t need, uint32_t dim);
hipError_t launch_bf16_scan(const float* d_Q, uint32_t n_queries, const ScanArgs& a, const uint16_t* d_mirror,
                            const float* d_norm2, const Bf16Scratch& s, uint32_t need, uint32_t k,
                            const float* d_vmax_sq, hipStream_t st);
double bf16_list_estimate(uint64_t n_rows, uint32_t need, uint32_t dim);
// most queries one launch_bf16_scan / launch_bf16_finish pair takes at this dimension
uint32_t bf16_max_batch(uint32_t dim, uint32_t metric, uint32_t n_queries);
bool bf16_uses_qs(uint32_t dim, uint32_t metric, uint32_t n_queries);
// Shortens every query's candidate list to the rows within 2 eps of its need-th largest approximate value
// (cand / ckeys hold rows and values side by side); updates thr[] and the control blocks.
void launch_qs_prune(uint32_t metric, const Bf16Scratch& s, uint32_t n_queries, uint32_t dim, hipStream_t st);
// query-stationary form of the prefilter (trag_qstat.hip)
bool qs_supports(uint32_t dim, uint32_t metric);
uint32_t qs_group_queries(uint32_t dim);
uint32_t qs_max_batch(uint32_t dim);
uint32_t qs_sample_step(uint64_t n_rows, uint32_t need, uint32_t dim);
hipError_t launch_qs_scan(uint32_t n_queries, const ScanArgs& a, const uint16_t* d_mirror,
                          const float* d_norm2, const Bf16Scratch& s, uint32_t need, hipStream_t st);
// bf16 mirror of the corpus (trag_mirror.hip) and the stages after its scan (trag_kernels.hip)
hipError_t launch_to_bf16(const float* d_src, uint64_t n_floats, uint16_t* d_dst, hipStream_t st);
bool mirror_scan_supports_dim(uint32_t dim);
hipError_t launch_scan_mirror(const uint16_t* d_mirror, const float* d_q, const float* d_norm2, const ScanArgs& a,
                              const Scratch& s, uint32_t need, uint32_t k, const float* d_vmax_sq,
                              uint32_t* n_waves, hipStream_t st);
hipError_t launch_select_mirror(const float* d_q, const ScanArgs& a, const Scratch& s, bool multi_level,
                                uint32_t live, uint32_t need, uint32_t k, uint32_t n_waves,
                                uint32_t* d_out_rows, float* d_out_scores, uint32_t* d_out_n,
                                uint32_t* d_status, hipStream_t st);
hipError_t launch_bf16_finish(const float* d_Q, uint32_t n_queries, const ScanArgs& a, const Bf16Scratch& s,
                              uint32_t live, uint32_t out_stride, uint32_t* d_out_rows, float* d_out_scores,
                              uint32_t* d_out_n, uint32_t* d_status, hipStream_t st);
// row i of the call is row row0 + i * row_step of the generator's stream
hipError_t launch_fill_synthetic(float* d_rows, uint64_t row0, uint64_t n, uint32_t dim,
                                 uint64_t seed, float* d_norm_tmp, hipStream_t st, uint64_t row_step = 1);
hipError_t launch_merge_topk(const uint32_t* d_rows, const float* d_scores, const uint32_t* d_list_n,
                             uint32_t n_lists, uint32_t k_in, uint32_t k_out, uint32_t* d_out_rows,
                             float* d_out_scores, uint32_t* d_out_n, hipStream_t st);
// global row = row0 + local row * row_mul; d_status (optional): per-query status of the local
// search, carried in bits 16..23 of the record's count word
hipError_t launch_pack_topk(const uint32_t* d_rows, const float* d_scores, const uint32_t* d_n,
                            uint32_t row0, uint32_t n_queries, uint32_t k, uint32_t* d_packed,
                            hipStream_t st, uint32_t row_mul = 1, const uint32_t* d_status = nullptr);
// d_out_status (optional): per query, the worst status any shard reported
hipError_t launch_merge_packed(const uint32_t* d_gathered, uint32_t n_lists, uint32_t n_queries,
                               uint32_t k, uint32_t* d_out_rows, float* d_out_scores,
                               uint32_t* d_out_n, hipStream_t st, uint32_t* d_out_status = nullptr);
hipError_t launch_zero_words(uint32_t* a, uint32_t na, uint32_t* b, uint32_t nb, hipStream_t st);
hipError_t launch_stream_probe(const float* d_rows, uint64_t n_floats, float* d_sink, hipStream_t st);
hipError_t launch_update_vmax(const float* d_rows, uint64_t n, uint32_t dim, float* d_vmax_sq,
                              float* d_norm2_out, hipStream_t st);

}  // namespace trag
