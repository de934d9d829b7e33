// The CPU side of a pinned-slot copy (devmem.hip: copy_in / copy_out): bytes between the caller's buffer and a pinned slot, over a
// few threads once a piece is large enough to be worth their start-up.  Host-only C++ on purpose: tests/native/host_copy_test.cpp
// includes it (the first version cut the piece into floor(n / 4)-sized parts and lost the last n % 4 bytes of a 6.2 MB upload).
#pragma once
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

namespace ugvc {

constexpr size_t kHostCopyParallelFrom = 4u << 20;     // pieces of at least this many bytes are split over threads

// the byte ranges [first, last) the `t` workers of an n-byte copy take: 64-byte aligned cuts, together exactly [0, n)
inline void host_copy_cuts(size_t n, unsigned t, std::vector<size_t>& cuts) {
    const size_t piece = ((n + t - 1) / t + 63) & ~(size_t)63;
    cuts.assign((size_t)t + 1, n);
    for (unsigned k = 0; k <= t; ++k) cuts[k] = std::min(n, (size_t)k * piece);
}

inline unsigned host_copy_threads(size_t n) {
    // (eight threads saturate one socket's copy bandwidth on the MI355X hosts: the 3.1 GB reference upload of the CLI is bound by
    // this copy, not by the link; a small host gets half its cores)
    return n >= kHostCopyParallelFrom ? std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency() / 2)) : 1u;
}

inline void host_copy(void* dst, const void* src, size_t n, unsigned threads = 0) {
    const unsigned t = threads ? threads : host_copy_threads(n);
    if (t <= 1) { memcpy(dst, src, n); return; }
    std::vector<size_t> cuts;
    host_copy_cuts(n, t, cuts);
    std::vector<std::thread> th;
    for (unsigned k = 1; k < t; ++k) {
        const size_t a = cuts[k], b = cuts[k + 1];
        if (b > a) th.emplace_back([=] { memcpy(static_cast<char*>(dst) + a, static_cast<const char*>(src) + a, b - a); });
    }
    memcpy(dst, src, cuts[1]);
    for (auto& x : th) x.join();
}

}  // namespace ugvc
