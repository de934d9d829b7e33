// tests/native/host_copy_test.cpp: csrc/host_copy.hpp - every byte of a copy arrives, nothing beyond it is written, for sizes around
// the threading threshold and around multiples of 64 x threads (the shape that lost a buffer's tail), with 1 ... 9 threads.
#include <stdint.h>
#include <stdio.h>

#include "../../variantcalling_amd/csrc/host_copy.hpp"

int main() {
    using namespace ugvc;
    long bad = 0, cases = 0;
    std::vector<size_t> sizes = {0, 1, 63, 64, 65, 4095, 6239490, 6239488, 6239491, (4u << 20) - 1, 4u << 20, (4u << 20) + 1, (4u << 20) + 255,
                                 (4u << 20) + 256, (4u << 20) + 257, 5000000, 8388608 + 3, 1559872 * 4 + 1, 1559872 * 4 + 2, 1559872 * 4 + 3};
    for (size_t n : sizes)
        for (unsigned t = 0; t <= 9; ++t) {
            std::vector<uint8_t> src(n + 64), dst(n + 64, 0xEE);
            for (size_t i = 0; i < src.size(); ++i) src[i] = (uint8_t)((i * 131u + (i >> 9)) & 0xFF);
            host_copy(dst.data(), src.data(), n, t);
            ++cases;
            for (size_t i = 0; i < n; ++i)
                if (dst[i] != src[i]) { ++bad; printf("n %zu threads %u: byte %zu not copied\n", n, t, i); break; }
            for (size_t i = n; i < n + 64; ++i)
                if (dst[i] != 0xEE) { ++bad; printf("n %zu threads %u: byte %zu beyond the copy written\n", n, t, i); break; }
            if (t >= 1) {
                std::vector<size_t> cuts;
                host_copy_cuts(n, t, cuts);
                if (cuts.front() != 0 || cuts.back() != n) { ++bad; printf("n %zu threads %u: cuts do not span the copy\n", n, t); }
                for (size_t k = 0; k + 1 < cuts.size(); ++k)
                    if (cuts[k] > cuts[k + 1] || (cuts[k] % 64 && cuts[k] != n)) { ++bad; printf("n %zu threads %u: bad cut %zu\n", n, t, cuts[k]); }
            }
        }
    printf("cases %ld bad %ld\n", cases, bad);
    return bad ? 1 : 0;
}
