// Device-wide primitives of the "next" rows (score evaluation, SEC database build): a stable LSD radix sort of
// (u64 key, u32 value) pairs and prefix sums, hand-written for gfx950 (kernels_prims.hip) - wave64 ballots for the
// in-wave digit ranking, LDS for the per-block histograms.  (Round 2 called hipCUB here; a CUB-shaped compatibility
// layer has no place in this library.)
#pragma once
#include "ugvc_device.hpp"

namespace ugvc {

// Stable ascending sort of n (key, value) pairs by the full 64-bit key.  k0 / v0 hold the input and are clobbered;
// k1 / v1 are scratch of the same size; *k_out / *v_out point at whichever pair of buffers holds the result.  Passes
// whose digit is the same for every key (the high bytes of contig << 32 | pos keys, of scores in a narrow range) are
// skipped.  Work buffers live in `tmp` (grown as needed).  Stream-ordered on ctx->stream except for one small
// read-back of the digit census.
int radix_sort_pairs_u64(ugvc_ctx* ctx, DeviceBuf& tmp, uint64_t* k0, uint64_t* k1, uint32_t* v0, uint32_t* v1, int64_t n,
                         uint64_t** k_out, uint32_t** v_out);

// In-place scans of n 64-bit words on ctx->stream (exclusive: out[i] = sum of in[0..i); inclusive: in[0..i]).
// Two 32-bit counters packed in one word scan together as long as neither sum reaches 2^32.
int scan_u64(ugvc_ctx* ctx, DeviceBuf& tmp, uint64_t* data, int64_t n, bool inclusive);

}  // namespace ugvc
