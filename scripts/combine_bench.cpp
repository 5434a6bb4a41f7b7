// scripts/combine_bench.cpp -- how well does mx_index_search combine concurrent single-query callers?
// Build: g++ -O2 -std=c++17 -pthread -I include scripts/combine_bench.cpp -L memex_amd -lmemex_hip \
//        -Wl,-rpath,$PWD/memex_amd -o build_ub/combine_bench
// Not product code: T threads each issue single-query mx_index_search calls (the reference's request
// pattern, api/handlers.rs:55-109) against one resident index; prints calls/s and queries per GPU batch.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include "memex_hip.h"
static inline float rnd(unsigned long long &s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return ((int)(s >> 40) - (1 << 23)) / (float)(1 << 23); }
int main(int argc, char **argv) {
    const size_t n = argc > 1 ? atoll(argv[1]) : 2000000; const int d = 384, k = 10;
    std::vector<float> x(n * d); unsigned long long s = 1;
    for (auto &v : x) v = rnd(s) + rnd(s) + rnd(s);
    mx_index *idx = nullptr;
    if (mx_index_open("bench", d, 0, &idx) != MX_OK || mx_index_add(idx, x.data(), n, nullptr) != MX_OK) { printf("setup: %s\n", mx_last_error()); return 1; }
    std::vector<float> q(4096 * d); for (auto &v : q) v = rnd(s) + rnd(s) + rnd(s);
    for (int threads : {1, 4, 16, 64, 256, 1024}) {
        const int per = threads <= 16 ? 200 : (threads <= 256 ? 64 : 16);
        mx_index_reset_stats(idx);
        std::atomic<int> bad{0};
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> ts;
        for (int t = 0; t < threads; ++t)
            ts.emplace_back([&, t] {
                uint64_t ids[k]; float sc[k]; int32_t nf;
                for (int j = 0; j < per; ++j)
                    if (mx_index_search(idx, q.data() + (size_t)((t * per + j) % 4096) * d, 1, k, ids, sc, nullptr, &nf) != MX_OK || nf != k) bad++;
            });
        for (auto &th : ts) th.join();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        mx_index_stats st; mx_index_get_stats(idx, &st);
        printf("%4d threads x %3d single-query calls on %zu x %d: %8.0f calls/s, %.1f queries per GPU batch, %d errors\n", threads, per, n, d, threads * per / dt, (double)st.queries / (double)st.searches, bad.load());
    }
    mx_index_close(idx);
    return 0;
}
