// The helper-thread hand-off of the sharded index (memex_amd/csrc/shard_pool.h) without a GPU: every job
// runs exactly once per helper and per round, run() returns only after all of them, both the spinning and
// the sleeping wake-up paths work, and the pool shuts down cleanly.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

#include "../../memex_amd/csrc/shard_pool.h"

int main() {
    int fails = 0;
    for (int helpers : {0, 1, 7}) {
        mx::ShardPool pool(helpers);
        std::vector<std::atomic<long>> hits(helpers + 1);
        for (auto &h : hits) h = 0;
        std::atomic<int> inside{0};
        const int rounds = 20000;
        for (int r = 0; r < rounds; ++r) {
            pool.run([&](int g) {
                inside.fetch_add(1);
                hits[g].fetch_add(1);
                if ((r % 5000) == 0 && g == helpers) std::this_thread::sleep_for(std::chrono::milliseconds(2));  // a straggler
                inside.fetch_sub(1);
            });
            if (inside.load() != 0) ++fails;           // run() returned while a job was still running
            for (int g = 0; g <= helpers; ++g)
                if (hits[g].load() != r + 1) ++fails;  // a job was skipped or ran twice
            if (r == 100 || r == 200) std::this_thread::sleep_for(std::chrono::milliseconds(5));  // helpers fall asleep
        }
        std::printf("helpers %d: %d rounds, fails %d\n", helpers, rounds, fails);
    }
    {   // results written by a helper are visible to the caller after run()
        mx::ShardPool pool(3);
        std::vector<int> out(4, 0);
        for (int r = 1; r <= 1000; ++r) {
            pool.run([&](int g) { out[g] = r * 10 + g; });
            for (int g = 0; g < 4; ++g)
                if (out[g] != r * 10 + g) ++fails;
        }
    }
    if (fails) {
        std::printf("FAILED (%d)\n", fails);
        return 1;
    }
    std::printf("OK shard pool\n");
    return 0;
}
