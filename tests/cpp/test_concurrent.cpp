// Host-entry searches of one handle from several threads (search(&self) is concurrent in the
// reference, src/index.rs:386): the same queries by 1 thread and by 4 threads, results compared
// bit for bit, throughput of both printed as one JSON line. C++ because Python threads would
// serialise on the interpreter lock around a call this short.
//   test_concurrent <rows> <dim> <queries per thread> <threads>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "trag.h"

int main(int argc, char** argv) {
    const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 10000;
    const uint32_t d = argc > 2 ? (uint32_t)atoi(argv[2]) : 384;
    const int per_thread = argc > 3 ? atoi(argv[3]) : 2000;
    const int n_threads = argc > 4 ? atoi(argv[4]) : 4;
    const uint32_t k = 10, NQ = 64;
    trag_config cfg{d, TRAG_COSINE, 0, TRAG_SCAN_AUTO, n};
    trag_store* s = nullptr;
    if (trag_store_create(&cfg, &s) != TRAG_OK) { fprintf(stderr, "create: %s\n", trag_last_error()); return 2; }
    if (trag_store_fill_synthetic(s, n, 1001, 0, nullptr) != TRAG_OK) { fprintf(stderr, "fill: %s\n", trag_last_error()); return 2; }
    // queries: stored rows with a perturbation that depends on the query only
    std::vector<float> Q((size_t)NQ * d);
    for (uint32_t b = 0; b < NQ; ++b) {
        if (trag_store_read_rows(s, (b * 131u) % n, 1, Q.data() + (size_t)b * d) != TRAG_OK) return 2;
        for (uint32_t j = 0; j < d; ++j) Q[(size_t)b * d + j] += 0.05f * (float)((b * 7u + j * 13u) % 11u) / 11.0f;
    }
    // the reference answers, one thread
    std::vector<uint32_t> ref_rows((size_t)NQ * k), ref_n(NQ);
    std::vector<float> ref_scores((size_t)NQ * k);
    for (uint32_t b = 0; b < NQ; ++b)
        if (trag_store_search(s, Q.data() + (size_t)b * d, d, k, ref_rows.data() + (size_t)b * k,
                              ref_scores.data() + (size_t)b * k, &ref_n[b]) != TRAG_OK) { fprintf(stderr, "search: %s\n", trag_last_error()); return 2; }
    std::atomic<int> bad{0};
    auto work = [&](int t, int count) {
        std::vector<uint32_t> rows(k);
        std::vector<float> scores(k);
        uint32_t cnt = 0;
        for (int i = 0; i < count; ++i) {
            const uint32_t b = (uint32_t)(t * 17 + i) % NQ;
            if (trag_store_search(s, Q.data() + (size_t)b * d, d, k, rows.data(), scores.data(), &cnt) != TRAG_OK) { bad++; continue; }
            if (cnt != ref_n[b] || memcmp(rows.data(), ref_rows.data() + (size_t)b * k, cnt * 4) ||
                memcmp(scores.data(), ref_scores.data() + (size_t)b * k, cnt * 4)) bad++;
        }
    };
    auto timed = [&](int threads) {
        std::vector<std::thread> th;
        const auto t0 = std::chrono::steady_clock::now();
        for (int t = 0; t < threads; ++t) th.emplace_back(work, t, per_thread);
        for (auto& x : th) x.join();
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return (double)threads * per_thread / sec;
    };
    work(0, 200);                       // warm-up
    const double q1 = timed(1);
    timed(n_threads);                   // (creates the spare contexts)
    const double qn = timed(n_threads);
    const double q1b = timed(1);
    printf("{\"rows\": %llu, \"dim\": %u, \"threads\": %d, \"qps_1_thread\": %.1f, \"qps_n_threads\": %.1f, "
           "\"qps_1_thread_again\": %.1f, \"speedup\": %.3f, \"mismatches\": %d}\n",
           (unsigned long long)n, d, n_threads, q1, qn, q1b, qn / (q1 > q1b ? q1 : q1b), bad.load());
    trag_store_destroy(s);
    return bad.load() ? 1 : 0;
}
