// shard_pool.h -- the persistent helper threads of a sharded index (mx_index_open_sharded).
//
// A search on G shards runs shard 0's pipeline on the calling thread and shards 1 .. G-1 on G-1 helper
// threads (each pipeline is host-synchronous: it ends with a wait on its device).  At 8 shards the whole
// scan of BASELINE configs[2] is ~0.25 ms per batch, so creating and joining G-1 std::threads per batch
// (round 2) is a first-order cost; the helpers are therefore created once per index and handed one job
// per batch.  Hand-off: an epoch counter the helpers spin on for up to ~100 us (under load a helper never
// sleeps between two batches), then a condition variable (an idle index costs no CPU).
// Plain C++17, no HIP: tests/cpp/test_shard_pool.cpp exercises it without a GPU.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace mx {

class ShardPool {
public:
    // n_helpers threads; run() executes fn(0) on the caller and fn(1) .. fn(n_helpers) on the helpers
    explicit ShardPool(int n_helpers) {
        for (int i = 0; i < n_helpers; ++i) th_.emplace_back([this, i] { loop(i + 1); });
    }
    ~ShardPool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_.store(true, std::memory_order_release);
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    ShardPool(const ShardPool &) = delete;
    ShardPool &operator=(const ShardPool &) = delete;

    int helpers() const { return (int)th_.size(); }

    // One job for every helper plus the caller; returns when all of them are done.  Not re-entrant: the
    // owner serialises calls (the sharded index holds its mutex).
    void run(const std::function<void(int)> &fn) {
        if (th_.empty()) {
            fn(0);
            return;
        }
        job_ = &fn;
        pending_.store((int)th_.size(), std::memory_order_release);
        {
            std::lock_guard<std::mutex> lk(m_);  // under the lock: a helper about to sleep re-checks the epoch
            epoch_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_all();
        fn(0);
        if (!spin_until([this] { return pending_.load(std::memory_order_acquire) == 0; })) {
            std::unique_lock<std::mutex> lk(m_);
            cv_done_.wait(lk, [this] { return pending_.load(std::memory_order_acquire) == 0; });
        }
        job_ = nullptr;
    }

private:
    template <class Pred>
    static bool spin_until(Pred done) {  // ~100 us of polling, then give up
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 1;; ++spins) {
            if (done()) return true;
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#endif
            if ((spins & 0xff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(100)) return false;
        }
    }

    void loop(int id) {
        uint32_t seen = 0;
        for (;;) {
            auto fresh = [&] { return epoch_.load(std::memory_order_acquire) != seen || stop_.load(std::memory_order_acquire); };
            if (!spin_until(fresh)) {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, fresh);
            }
            if (stop_.load(std::memory_order_acquire)) return;
            seen = epoch_.load(std::memory_order_acquire);
            (*job_)(id);
            if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> lk(m_);
                cv_done_.notify_one();
            }
        }
    }

    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, cv_done_;
    std::atomic<uint32_t> epoch_{0};
    std::atomic<int> pending_{0};
    std::atomic<bool> stop_{false};
    const std::function<void(int)> *job_ = nullptr;
};

}  // namespace mx
