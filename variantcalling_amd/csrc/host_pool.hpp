// Host thread pool of the chunk pipeline (pipeline.hip).  Plain C++: tests/test_host_pool.py builds a stress test of it with g++.
#pragma once
#include <sched.h>
#include <stdint.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace ugvc {

// ---- a persistent pool: parallel_for(n_tasks, f) runs f(task) on the workers and on the caller.
// Jobs here are 50-200 us long and follow each other at once, so inside a call (burst(true) ... burst(false)) the workers
// poll for the next job instead of sleeping on the condition variable: a futex wake-up costs 30-100 us per job, which was
// half of a chunk's packing time (UGVC_PIPE_TRACE).  Tasks are claimed through ONE 64-bit ticket = generation << 32 | next
// task, so a worker still leaving the previous job can neither take nor repeat a task of the next one
// (tests/native/host_pool_stress.cpp).
class HostPool {
  public:
    // `cpus`: the workers stay on these CPUs (the GPU's NUMA node: staging buffers and copy engines are local to it)
    explicit HostPool(int n_threads, const cpu_set_t* cpus = nullptr) {
        if (cpus) { cpus_ = *cpus; pinned_ = true; }
        for (int t = 0; t < n_threads; ++t)
            th_.emplace_back([this] {
                if (pinned_) (void)sched_setaffinity(0, sizeof(cpus_), &cpus_);
                work();
            });
    }
    ~HostPool() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_.store(true);
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    void burst(bool on) {
        {
            std::lock_guard<std::mutex> g(m_);
            burst_.store(on);
        }
        if (on) cv_.notify_all();
    }
    void parallel_for(int n_tasks, const std::function<void(int)>& f) {
        if (n_tasks <= 0) return;
        // generations: even = open for claims, odd = closed.  The ticket is closed BEFORE the job fields change - a worker
        // still leaving the previous job holds a stale ticket value, and its claim must fail rather than succeed against
        // the new task count - and opened with the next even generation once they are in place.
        const uint64_t g = ticket_.load() >> 32;
        ticket_.store(((g + 1) << 32) | 0xffffffffu);
        job_ = &f;
        n_tasks_.store(n_tasks);
        pending_.store(n_tasks);
        const uint64_t gen = g + 2;
        {
            std::lock_guard<std::mutex> lk(m_);              // (a worker between its check and its wait must not miss this)
            ticket_.store(gen << 32);
        }
        if (!burst_.load()) cv_.notify_all();
        drain(gen);                                          // the caller works too
        for (unsigned spins = 0; pending_.load(std::memory_order_acquire) != 0;)
            relax(spins);
    }
    int size() const { return (int)th_.size() + 1; }

  private:
    // polling step: a few pause instructions, then the CPU is offered to whoever else is runnable on it - on a host with
    // fewer free cores than pool threads a worker that has claimed a task must not wait a scheduler quantum behind pollers
    static void relax(unsigned& spins) {
        if ((++spins & 63u) == 0) sched_yield();
        else __builtin_ia32_pause();
    }
    void drain(uint64_t gen) {
        for (;;) {
            uint64_t cur = ticket_.load(std::memory_order_acquire);
            if ((cur >> 32) != gen) return;
            const int t = (int)(cur & 0xffffffffu);
            if (t >= n_tasks_.load()) return;
            if (!ticket_.compare_exchange_weak(cur, cur + 1, std::memory_order_acq_rel)) continue;
            (*job_)(t);
            pending_.fetch_sub(1, std::memory_order_acq_rel);
        }
    }
    void work() {
        uint64_t seen = 0;
        unsigned spins = 0;
        for (;;) {
            uint64_t gen = ticket_.load(std::memory_order_acquire) >> 32;
            if (gen == seen || (gen & 1)) {
                if (stop_.load()) return;
                if (burst_.load()) {
                    relax(spins);
                    continue;
                }
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] {
                    const uint64_t now = ticket_.load() >> 32;
                    return stop_.load() || burst_.load() || (now != seen && !(now & 1));
                });
                continue;
            }
            seen = gen;
            spins = 0;
            drain(gen);
        }
    }
    cpu_set_t cpus_;
    bool pinned_ = false;
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_;
    const std::function<void(int)>* job_ = nullptr;
    std::atomic<uint64_t> ticket_{0};
    std::atomic<int> n_tasks_{0}, pending_{0};
    std::atomic<bool> burst_{false}, stop_{false};
};

}  // namespace ugvc
