/*
 * trag_oracle.c -- TEST INFRASTRUCTURE ONLY. Not part of the product.
 *
 * A plain-C restatement of the CPU arithmetic of trueno-rag's dense retrieval
 * hot path, used as the checker for the HIP implementation and as the timed
 * "port" CPU baseline. Nothing in the shipped library links or loads this file.
 *
 * Parity status: the reference is Rust and cannot be built in this image, and
 * it ships no golden-vector files. This restatement is pinned only by the
 * reference's inline known-answer unit tests (toy inputs: orderings, result
 * counts, 2/61), transcribed in tests/golden/ref_unit_vectors.json.
 *
 * Build:  gcc -O2 -ffp-contract=off -fno-fast-math  (never -ffast-math / FMA
 * contraction: the reference's sums are strict left-to-right f32, one rounding
 * for the multiply and one for the add).
 *
 * Reference lines followed (paths relative to /root/reference):
 *   src/index.rs:440-450   cosine_similarity   -> oracle_cosine
 *   src/index.rs:452-458   euclidean_distance  -> oracle_euclidean
 *   src/index.rs:460-462   dot_product         -> oracle_dot
 *   src/index.rs:386-412   VectorStore::search -> oracle_search / oracle_search_faithful
 *   src/fusion.rs:68-84    reciprocal_rank_fusion -> oracle_rrf
 *   src/fusion.rs:227-231  sort_by_score       -> oracle_rrf (ordering)
 *   src/fusion.rs:87-224   other fusion variants -> oracle_fuse
 *   crates/trueno-rag-cli/src/main.rs:480-492  index-ordered loop -> tie rule
 *
 * Tie rule. The reference iterates a HashMap with a per-process random hash
 * and then stable-sorts, so equal scores come out in random order. The
 * deterministic rule used here (and by the HIP path) is: equal scores are
 * ordered by ascending row index (= insertion order, what the CLI's
 * Vec-ordered copy of the same loop produces). NaN scores compare Equal to
 * everything under the reference's comparator, which makes its sort result
 * unspecified; here NaN scores are placed after every number, by ascending row.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_COSINE 0
#define ORACLE_EUCLIDEAN 1
#define ORACLE_DOT 2

/* src/index.rs:460-462: a.iter().zip(b).map(|(x,y)| x*y).sum()  (fold from 0.0) */
float oracle_dot(const float *a, const float *b, uint32_t d) {
    float s = 0.0f;
    for (uint32_t i = 0; i < d; ++i) {
        float p = a[i] * b[i];
        s = s + p;
    }
    return s;
}

/* src/index.rs:440-450 */
float oracle_cosine(const float *a, const float *b, uint32_t d) {
    float dot = 0.0f, na = 0.0f, nb = 0.0f;
    for (uint32_t i = 0; i < d; ++i) { float p = a[i] * b[i]; dot = dot + p; }
    for (uint32_t i = 0; i < d; ++i) { float p = a[i] * a[i]; na = na + p; }
    for (uint32_t i = 0; i < d; ++i) { float p = b[i] * b[i]; nb = nb + p; }
    na = sqrtf(na);
    nb = sqrtf(nb);
    if (na == 0.0f || nb == 0.0f) return 0.0f;
    float den = na * nb;
    return dot / den;
}

/* src/index.rs:452-458: sum of (x-y).powi(2) (= t*t), then sqrt */
float oracle_euclidean(const float *a, const float *b, uint32_t d) {
    float s = 0.0f;
    for (uint32_t i = 0; i < d; ++i) {
        float t = a[i] - b[i];
        float p = t * t;
        s = s + p;
    }
    return sqrtf(s);
}

/* src/index.rs:398-402: the score the store ranks by (higher is better). */
float oracle_score(int metric, const float *q, const float *v, uint32_t d) {
    switch (metric) {
    case ORACLE_COSINE: return oracle_cosine(q, v, d);
    case ORACLE_EUCLIDEAN: return -oracle_euclidean(q, v, d);
    default: return oracle_dot(q, v, d);
    }
}

/* Total order used for ranking: returns 1 when (sa,ra) ranks before (sb,rb). */
static inline int ranks_before(float sa, uint32_t ra, float sb, uint32_t rb) {
    int na = sa != sa, nb = sb != sb;
    if (na || nb) {
        if (na != nb) return nb; /* number before NaN */
        return ra < rb;
    }
    if (sa > sb) return 1;
    if (sa < sb) return 0;
    return ra < rb; /* equal (incl. -0 == +0): ascending row */
}

/* Scores of every row; dead[r] != 0 marks a removed row (skipped by search). */
void oracle_score_all(const float *corpus, uint64_t n, uint32_t d, int metric,
                      const float *q, float *out_scores) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < (int64_t)n; ++r)
        out_scores[r] = oracle_score(metric, q, corpus + (size_t)r * d, d);
}

/* Top-k by the total order, O(n*k) insertion for small k, used by oracle_search. */
static uint32_t select_topk(const float *scores, const uint8_t *dead, uint64_t n,
                            uint32_t k, uint32_t *out_rows, float *out_scores) {
    uint32_t m = 0;
    if (k == 0) return 0;
    for (uint64_t r = 0; r < n; ++r) {
        if (dead && dead[r]) continue;
        float s = scores[r];
        if (m == k && !ranks_before(s, (uint32_t)r, out_scores[m - 1], out_rows[m - 1]))
            continue;
        /* binary search for insertion point */
        uint32_t lo = 0, hi = m;
        while (lo < hi) {
            uint32_t mid = (lo + hi) / 2;
            if (ranks_before(out_scores[mid], out_rows[mid], s, (uint32_t)r)) lo = mid + 1;
            else hi = mid;
        }
        uint32_t end = (m < k) ? m : k - 1;
        for (uint32_t j = end; j > lo; --j) {
            out_scores[j] = out_scores[j - 1];
            out_rows[j] = out_rows[j - 1];
        }
        out_scores[lo] = s;
        out_rows[lo] = (uint32_t)r;
        if (m < k) ++m;
    }
    return m;
}

/*
 * src/index.rs:386-412 with a contiguous corpus and heap selection: same
 * arithmetic per row, same final order (modulo the tie rule above).
 * Returns the number of results = min(k, live rows).
 */
uint32_t oracle_search(const float *corpus, const uint8_t *dead, uint64_t n, uint32_t d,
                       int metric, const float *q, uint32_t k, uint32_t *out_rows,
                       float *out_scores) {
    if (k == 0 || n == 0) return 0;
    float *scores = (float *)malloc(sizeof(float) * (size_t)n);
    if (!scores) return 0;
    oracle_score_all(corpus, n, d, metric, q, scores);
    uint32_t m = select_topk(scores, dead, n, k, out_rows, out_scores);
    free(scores);
    return m;
}

/* ---- reference-faithful single-thread variant (the timed CPU baseline) ---- */

typedef struct {
    uint64_t id_hi, id_lo; /* 16-byte id, like ChunkId(Uuid) */
    float score;
} scored_t; /* 20 bytes of payload, as (ChunkId, f32) */

static void merge_sort_desc(scored_t *a, scored_t *tmp, size_t n) {
    /* stable bottom-up merge sort, comparator b.score.partial_cmp(a.score) with
     * NaN treated as Equal (src/index.rs:408) */
    for (size_t w = 1; w < n; w *= 2) {
        for (size_t lo = 0; lo < n; lo += 2 * w) {
            size_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            size_t i = lo, j = mid, o = lo;
            while (i < mid && j < hi) {
                /* take right only if strictly greater (stable) */
                if (a[j].score > a[i].score) tmp[o++] = a[j++];
                else tmp[o++] = a[i++];
            }
            while (i < mid) tmp[o++] = a[i++];
            while (j < hi) tmp[o++] = a[j++];
        }
        memcpy(a, tmp, n * sizeof(scored_t));
    }
}

/*
 * The reference's data layout and algorithm shape: every row is its own heap
 * allocation (Vec<f32>), rows are visited through a pointer table in `order`
 * (stand-in for HashMap iteration order; pass the identity for index order),
 * the query norm is recomputed per row (cosine_similarity does), every
 * (id, score) is materialised, the whole vector is stable-sorted, then
 * truncated to k.  row_ptrs[i] points at row i's d floats.
 */
uint32_t oracle_search_faithful(const float *const *row_ptrs, const uint32_t *order,
                                uint64_t n, uint32_t d, int metric, const float *q,
                                uint32_t k, uint32_t *out_rows, float *out_scores) {
    scored_t *v = (scored_t *)malloc(sizeof(scored_t) * (size_t)(n ? n : 1));
    scored_t *tmp = (scored_t *)malloc(sizeof(scored_t) * (size_t)(n ? n : 1));
    if (!v || !tmp) { free(v); free(tmp); return 0; }
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t r = order ? order[i] : (uint32_t)i;
        v[i].id_hi = 0;
        v[i].id_lo = r;
        v[i].score = oracle_score(metric, q, row_ptrs[r], d);
    }
    merge_sort_desc(v, tmp, (size_t)n);
    uint32_t m = (uint32_t)(n < k ? n : k);
    for (uint32_t i = 0; i < m; ++i) {
        out_rows[i] = (uint32_t)v[i].id_lo;
        out_scores[i] = v[i].score;
    }
    free(v);
    free(tmp);
    return m;
}

/* ------------------------------- fusion ---------------------------------- */

typedef struct {
    uint64_t id;
    float score;
    uint32_t first_seen; /* order of first appearance: dense list first, then sparse */
} fused_t;

static int find_id(const fused_t *t, uint32_t n, uint64_t id) {
    for (uint32_t i = 0; i < n; ++i)
        if (t[i].id == id) return (int)i;
    return -1;
}

/* order: score descending; ties by first appearance (dense list order, then
 * sparse list order); NaN last. The reference's tie order is hash order. */
static void sort_fused(fused_t *t, uint32_t n) {
    for (uint32_t i = 1; i < n; ++i) { /* insertion sort: n <= nd+ns, stable */
        fused_t x = t[i];
        uint32_t j = i;
        while (j > 0 && ranks_before(x.score, x.first_seen, t[j - 1].score, t[j - 1].first_seen)) {
            t[j] = t[j - 1];
            --j;
        }
        t[j] = x;
    }
}

/*
 * src/fusion.rs:68-84 + 227-231.  score[id] += 1.0/(k + rank as f32 + 1.0),
 * dense pass then sparse pass, f32 accumulate from 0.0; input scores ignored.
 * Returns the number of distinct ids written (<= nd+ns).
 */
uint32_t oracle_rrf(const uint64_t *dense_ids, uint32_t nd, const uint64_t *sparse_ids,
                    uint32_t ns, float k, uint64_t *out_ids, float *out_scores) {
    uint32_t cap = nd + ns, n = 0;
    fused_t *t = (fused_t *)malloc(sizeof(fused_t) * (cap ? cap : 1));
    for (uint32_t pass = 0; pass < 2; ++pass) {
        const uint64_t *ids = pass ? sparse_ids : dense_ids;
        uint32_t cnt = pass ? ns : nd;
        for (uint32_t rank = 0; rank < cnt; ++rank) {
            float den = k + (float)rank; /* (k + rank as f32) */
            den = den + 1.0f;            /* + 1.0 */
            float inc = 1.0f / den;
            int at = find_id(t, n, ids[rank]);
            if (at < 0) {
                t[n].id = ids[rank];
                t[n].score = 0.0f;
                t[n].first_seen = n;
                at = (int)n++;
            }
            t[at].score = t[at].score + inc;
        }
    }
    sort_fused(t, n);
    for (uint32_t i = 0; i < n; ++i) { out_ids[i] = t[i].id; out_scores[i] = t[i].score; }
    free(t);
    return n;
}

/* src/fusion.rs:183-202 */
static void min_max_normalize(const float *s, uint32_t n, float *out) {
    if (n == 0) return;
    float mn = INFINITY, mx = -INFINITY;
    for (uint32_t i = 0; i < n; ++i) { mn = fminf(mn, s[i]); mx = fmaxf(mx, s[i]); }
    float range = mx - mn;
    if (fabsf(range) < 1.1920929e-07f) { for (uint32_t i = 0; i < n; ++i) out[i] = 1.0f; return; }
    for (uint32_t i = 0; i < n; ++i) out[i] = (s[i] - mn) / range;
}

/* src/fusion.rs:205-224 (population variance, /n) */
static void z_score_normalize(const float *s, uint32_t n, float *out) {
    if (n == 0) return;
    float fn = (float)n, sum = 0.0f;
    for (uint32_t i = 0; i < n; ++i) sum = sum + s[i];
    float mean = sum / fn, var = 0.0f;
    for (uint32_t i = 0; i < n; ++i) { float t = s[i] - mean; var = var + t * t; }
    var = var / fn;
    float sd = sqrtf(var);
    if (fabsf(sd) < 1.1920929e-07f) { for (uint32_t i = 0; i < n; ++i) out[i] = 0.0f; return; }
    for (uint32_t i = 0; i < n; ++i) out[i] = (s[i] - mean) / sd;
}

#define ORACLE_FUSE_RRF 0
#define ORACLE_FUSE_LINEAR 1
#define ORACLE_FUSE_CONVEX 2
#define ORACLE_FUSE_DBSF 3
#define ORACLE_FUSE_UNION 4
#define ORACLE_FUSE_INTERSECTION 5

/*
 * src/fusion.rs:42-63 dispatch over all six variants. `param` is k (RRF),
 * dense_weight (Linear) or alpha (Convex). Duplicate ids inside one input
 * list follow the reference's HashMap semantics (accumulate for RRF/Linear/
 * DBSF; last one wins for Union's dense pass and Intersection's maps).
 */
uint32_t oracle_fuse(int strategy, float param, const uint64_t *dense_ids,
                     const float *dense_scores, uint32_t nd, const uint64_t *sparse_ids,
                     const float *sparse_scores, uint32_t ns, uint64_t *out_ids,
                     float *out_scores) {
    uint32_t cap = nd + ns, n = 0;
    if (strategy == ORACLE_FUSE_RRF)
        return oracle_rrf(dense_ids, nd, sparse_ids, ns, param, out_ids, out_scores);
    fused_t *t = (fused_t *)malloc(sizeof(fused_t) * (cap ? cap : 1));
    float *dn = (float *)malloc(sizeof(float) * (nd ? nd : 1));
    float *sn = (float *)malloc(sizeof(float) * (ns ? ns : 1));
    if (strategy == ORACLE_FUSE_LINEAR || strategy == ORACLE_FUSE_CONVEX ||
        strategy == ORACLE_FUSE_DBSF) {
        float wd = 1.0f, ws = 1.0f;
        if (strategy == ORACLE_FUSE_DBSF) {
            z_score_normalize(dense_scores, nd, dn);
            z_score_normalize(sparse_scores, ns, sn);
        } else {
            wd = param;
            ws = 1.0f - param;
            min_max_normalize(dense_scores, nd, dn);
            min_max_normalize(sparse_scores, ns, sn);
        }
        for (uint32_t pass = 0; pass < 2; ++pass) {
            const uint64_t *ids = pass ? sparse_ids : dense_ids;
            const float *nrm = pass ? sn : dn;
            uint32_t cnt = pass ? ns : nd;
            float w = pass ? ws : wd;
            for (uint32_t i = 0; i < cnt; ++i) {
                int at = find_id(t, n, ids[i]);
                if (at < 0) { t[n].id = ids[i]; t[n].score = 0.0f; t[n].first_seen = n; at = (int)n++; }
                float inc = (strategy == ORACLE_FUSE_DBSF) ? nrm[i] : w * nrm[i];
                t[at].score = t[at].score + inc;
            }
        }
        sort_fused(t, n);
    } else if (strategy == ORACLE_FUSE_UNION) {
        /* src/fusion.rs:141-160: dense (score, rank) inserted (later duplicate
         * overwrites), sparse only if absent with rank nd+rank; sort by rank. */
        uint32_t *rank = (uint32_t *)malloc(sizeof(uint32_t) * (cap ? cap : 1));
        for (uint32_t i = 0; i < nd; ++i) {
            int at = find_id(t, n, dense_ids[i]);
            if (at < 0) { t[n].id = dense_ids[i]; at = (int)n++; }
            t[at].score = dense_scores[i];
            rank[at] = i;
        }
        for (uint32_t i = 0; i < ns; ++i) {
            int at = find_id(t, n, sparse_ids[i]);
            if (at < 0) { t[n].id = sparse_ids[i]; t[n].score = sparse_scores[i]; rank[n] = nd + i; n++; }
        }
        for (uint32_t i = 1; i < n; ++i) { /* ranks are unique: plain insertion sort */
            fused_t x = t[i]; uint32_t rx = rank[i], j = i;
            while (j > 0 && rank[j - 1] > rx) { t[j] = t[j - 1]; rank[j] = rank[j - 1]; --j; }
            t[j] = x; rank[j] = rx;
        }
        free(rank);
    } else { /* ORACLE_FUSE_INTERSECTION, src/fusion.rs:163-180 */
        for (uint32_t i = 0; i < nd; ++i) {
            /* HashMap collect: the last duplicate of an id wins in each list */
            int last_d = -1, last_s = -1;
            if (find_id(t, n, dense_ids[i]) >= 0) continue; /* id already emitted */
            for (uint32_t j = 0; j < nd; ++j) if (dense_ids[j] == dense_ids[i]) last_d = (int)j;
            for (uint32_t j = 0; j < ns; ++j) if (sparse_ids[j] == dense_ids[i]) last_s = (int)j;
            if (last_s < 0) continue;
            t[n].id = dense_ids[i];
            t[n].score = (dense_scores[last_d] + sparse_scores[last_s]) / 2.0f;
            t[n].first_seen = n;
            n++;
        }
        sort_fused(t, n);
    }
    for (uint32_t i = 0; i < n; ++i) { out_ids[i] = t[i].id; out_scores[i] = t[i].score; }
    free(t); free(dn); free(sn);
    return n;
}

/* ----------------- synthetic corpus generator (host side) ----------------- */
/*
 * Counter-based generator keyed by (seed, row, col): every component is the
 * sum of four 16-bit uniforms (Irwin-Hall, approximately normal) taken from a
 * splitmix64 hash, centred and scaled; integer arithmetic only until one exact
 * int->float conversion and one multiply by a power-of-two-free constant, so
 * host and device produce identical bits. Rows are then L2-normalised with the
 * sequential f32 sum (as MockEmbedder normalises, src/embed.rs:147-154).
 */
static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

float oracle_synth_component(uint64_t seed, uint64_t row, uint32_t col) {
    uint64_t h = splitmix64(seed ^ splitmix64(row * 0x100000001B3ull + col));
    int32_t s = (int32_t)(h & 0xFFFF) + (int32_t)((h >> 16) & 0xFFFF) +
                (int32_t)((h >> 32) & 0xFFFF) + (int32_t)((h >> 48) & 0xFFFF) - 131070;
    return (float)s * (1.0f / 37837.0f); /* sd of the 4-uniform sum = 65536/sqrt(3) */
}

void oracle_synth_rows(uint64_t seed, uint64_t row0, uint64_t n, uint32_t d, int normalize,
                       float *out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        float *v = out + (size_t)i * d;
        for (uint32_t c = 0; c < d; ++c) v[c] = oracle_synth_component(seed, row0 + (uint64_t)i, c);
        if (normalize) {
            float ss = 0.0f;
            for (uint32_t c = 0; c < d; ++c) { float p = v[c] * v[c]; ss = ss + p; }
            float nrm = sqrtf(ss);
            if (nrm > 0.0f)
                for (uint32_t c = 0; c < d; ++c) v[c] = v[c] / nrm;
        }
    }
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------- timed CPU baseline ---------------------------- */
#include <time.h>

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/*
 * The reference's CPU path restated and timed on one thread: every row is
 * copied into its own heap allocation (Vec<f32>), rows are visited in a
 * scrambled order (stand-in for HashMap iteration, src/index.rs:394-397), and
 * each query runs oracle_search_faithful (score all, stable sort all, truncate).
 * Only the searches are timed (the store is built beforehand). Returns the
 * median seconds per query; the last query's results go to out_rows/out_scores.
 */
double oracle_bench_faithful(const float *corpus, uint64_t n, uint32_t d, int metric,
                             const float *queries, uint32_t nq, uint32_t k, uint32_t *out_rows,
                             float *out_scores) {
    float **rows = (float **)malloc(sizeof(float *) * (size_t)(n ? n : 1));
    uint32_t *order = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n ? n : 1));
    double *t = (double *)malloc(sizeof(double) * (nq ? nq : 1));
    for (uint64_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
    for (uint64_t i = n; i > 1; --i) { /* Fisher-Yates with the integer hash */
        uint64_t j = splitmix64(i * 0x9E3779B97F4A7C15ull) % i;
        uint32_t tmp = order[i - 1]; order[i - 1] = order[j]; order[j] = tmp;
    }
    for (uint64_t i = 0; i < n; ++i) { /* allocate in visiting order, like inserts into a map */
        uint32_t r = order[i];
        rows[r] = (float *)malloc(sizeof(float) * d);
        memcpy(rows[r], corpus + (size_t)r * d, sizeof(float) * d);
    }
    for (uint32_t qi = 0; qi < nq; ++qi) {
        double t0 = now_s();
        oracle_search_faithful((const float *const *)rows, order, n, d, metric,
                               queries + (size_t)qi * d, k, out_rows, out_scores);
        t[qi] = now_s() - t0;
    }
    for (uint32_t i = 1; i < nq; ++i) { /* sort times */
        double x = t[i]; uint32_t j = i;
        while (j > 0 && t[j - 1] > x) { t[j] = t[j - 1]; --j; }
        t[j] = x;
    }
    double med = nq ? t[nq / 2] : 0.0;
    for (uint64_t i = 0; i < n; ++i) free(rows[i]);
    free(rows); free(order); free(t);
    return med;
}
