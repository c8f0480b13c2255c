// mm_pool.h -- a small persistent worker pool for the host-side point transforms (chain walk,
// between-pullback moves, refinement-grid rebuild).  std::thread creation costs ~30 us per thread,
// which is as much as the work of one chunk here; the pool's workers sleep on a condition variable
// between calls.  parallel_for(n, fn) runs fn(0..n-1) on the workers and the calling thread and
// returns when all are done.  One job at a time (calls are serialised by a mutex); results do not
// depend on the number of workers because every index writes disjoint data.
#pragma once

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cstdlib>
#include <thread>
#include <vector>

namespace mm {

class WorkerPool {
public:
    static WorkerPool& instance()
    {
        static WorkerPool pool;
        return pool;
    }

    int workers() const { return (int)threads_.size() + 1; }

    void parallel_for(int n, const std::function<void(int)>& fn)
    {
        if (n <= 0) return;
        if (n == 1 || threads_.empty()) { for (int i = 0; i < n; ++i) fn(i); return; }
        std::lock_guard<std::mutex> serial(call_mu_);
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn; n_ = n; next_.store(0); pending_ = (int)threads_.size(); ++epoch_;
        }
        cv_.notify_all();
        run_indices();
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }

    WorkerPool(const WorkerPool&) = delete;
    WorkerPool& operator=(const WorkerPool&) = delete;

private:
    WorkerPool()
    {
        unsigned hw = std::thread::hardware_concurrency();
        // one process per GPU (torchrun exports LOCAL_WORLD_SIZE): the ranks of a node share its cores
        if (const char* lws = std::getenv("LOCAL_WORLD_SIZE")) {
            const int k = std::atoi(lws);
            if (k > 1 && hw) hw = std::max(2u, hw / (unsigned)k);
        }
        const int nw = (int)std::max(1u, std::min(16u, hw ? hw : 1u)) - 1;   // a GPU box's CPU share is 16 cores
        for (int i = 0; i < nw; ++i) threads_.emplace_back([this] { loop(); });
    }
    ~WorkerPool()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (std::thread& t : threads_) t.join();
    }
    void run_indices()
    {
        for (;;) {
            const int i = next_.fetch_add(1);
            if (i >= n_) break;
            (*fn_)(i);
        }
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || epoch_ != seen; });
                if (stop_) return;
                seen = epoch_;
            }
            run_indices();
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (--pending_ == 0) done_cv_.notify_one();
            }
        }
    }

    std::vector<std::thread> threads_;
    std::mutex mu_, call_mu_;
    std::condition_variable cv_, done_cv_;
    const std::function<void(int)>* fn_ = nullptr;
    int n_ = 0, pending_ = 0;
    std::atomic<int> next_{0};
    uint64_t epoch_ = 0;
    bool stop_ = false;
};

inline void parallel_for(int n, const std::function<void(int)>& fn) { WorkerPool::instance().parallel_for(n, fn); }

}  // namespace mm
