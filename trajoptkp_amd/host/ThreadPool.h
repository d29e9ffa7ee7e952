// ThreadPool.h -- persistent worker pool for the host finite-difference harness.
// The reference spawns and joins hardware_concurrency()-1 std::threads on EVERY derivative call
// (src/Optimiser/Optimiser.cpp:217-236,239-323); here the workers live as long as the Differentiator and
// pull items off an atomic counter, so a call costs one wake-up instead of a thread create/join per worker.
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

class ThreadPool {
public:
    explicit ThreadPool(int nthreads) : n_(nthreads < 1 ? 1 : nthreads)
    {
        for (int i = 1; i < n_; i++) workers_.emplace_back([this, i] { loop(i); });
    }
    ~ThreadPool()
    {
        { std::lock_guard<std::mutex> g(mu_); stop_ = true; generation_++; }
        cv_.notify_all();
        for (std::thread &t : workers_) t.join();
    }
    int size() const { return n_; }

    // fn(item, tid) for item = 0..count-1, dynamic scheduling; the caller is worker 0.  Returns when every
    // item is done.  Not re-entrant.
    void parallel_for(int count, const std::function<void(int, int)> &fn)
    {
        if (count <= 0) return;
        if (n_ == 1 || count == 1) { for (int i = 0; i < count; i++) fn(i, 0); return; }
        {
            std::lock_guard<std::mutex> g(mu_);
            fn_ = &fn; count_ = count; next_.store(0); pending_ = n_ - 1; generation_++;
        }
        cv_.notify_all();
        run(0);
        std::unique_lock<std::mutex> g(mu_);
        done_.wait(g, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }

private:
    void run(int tid)
    {
        for (;;) {
            const int i = next_.fetch_add(1);
            if (i >= count_) break;
            (*fn_)(i, tid);
        }
    }
    void loop(int tid)
    {
        unsigned long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&] { return generation_ != seen; });
                seen = generation_;
                if (stop_) return;
            }
            run(tid);
            {
                std::lock_guard<std::mutex> g(mu_);
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
    int n_;
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    const std::function<void(int, int)> *fn_ = nullptr;
    std::atomic<int> next_{0};
    int count_ = 0, pending_ = 0;
    unsigned long generation_ = 0;
    bool stop_ = false;
};
