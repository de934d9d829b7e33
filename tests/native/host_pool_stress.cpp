// TEST INFRASTRUCTURE: stress of csrc/host_pool.hpp (built by tests/test_host_pool.py with g++ -pthread).
// Thousands of back-to-back jobs of tiny tasks, in and out of burst mode, with task counts that grow and shrink from job to
// job (the case in which a worker still leaving job k could take or repeat a task of job k + 1): every task of every job
// must run exactly once and parallel_for must not return before the last one has.
#include <stdio.h>
#include <stdlib.h>

#include "../../variantcalling_amd/csrc/host_pool.hpp"

int main(int argc, char** argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 8, jobs = argc > 2 ? atoi(argv[2]) : 20000;
    ugvc::HostPool pool(threads - 1);
    std::vector<std::atomic<int>> hits(64);
    long bad = 0, total = 0;
    unsigned lcg = 12345;
    for (int j = 0; j < jobs; ++j) {
        if (j % 1000 == 0) pool.burst((j / 1000) % 2 == 0);
        lcg = lcg * 1664525u + 1013904223u;
        const int n = 1 + (int)((lcg >> 16) % 64);
        for (auto& h : hits) h.store(0);
        std::atomic<int> done{0};
        const std::function<void(int)> f = [&](int t) {
            hits[(size_t)t].fetch_add(1);
            done.fetch_add(1);
        };
        pool.parallel_for(n, f);
        if (done.load() != n) ++bad;
        for (int t = 0; t < 64; ++t)
            if (hits[(size_t)t].load() != (t < n ? 1 : 0)) ++bad;
        total += n;
    }
    pool.burst(false);
    printf("jobs %d tasks %ld bad %ld\n", jobs, total, bad);
    return bad ? 1 : 0;
}
