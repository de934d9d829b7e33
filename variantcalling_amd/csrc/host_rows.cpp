// Host-only row checks of the variant columns (what the kernels rely on: contig range, non-empty alleles inside the pool,
// POS >= 1, rows sorted by (contig, pos)) - the replacement of the reference's per-record loop needs them at memory speed:
// a scalar loop with an early exit costs ~5 ns a row, as much as copying the row's 39 bytes.  Here blocks of rows are
// checked branch-free, column group by column group, so that the compiler vectorises them (function multiversioning: the
// AVX-512 / AVX2 clone is picked at load time); only a block that holds an offending row is re-read row by row to name it.
// Compiled by g++ (see Makefile): no HIP in this file.
#include <immintrin.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>

#include "../../include/ugvc_mi355x.h"

namespace ugvc {

__attribute__((target_clones("avx512f", "avx2", "default")))
static unsigned block_bad(const uint16_t* __restrict__ ct, const int32_t* __restrict__ pos, const uint16_t* __restrict__ rl,
                          const uint16_t* __restrict__ al, const uint32_t* __restrict__ ro, const uint32_t* __restrict__ ao,
                          int64_t lo, int64_t hi, unsigned n_contigs, uint64_t alleles_len, int64_t* n_indel) {
    unsigned bad = 0, k = 0;
    const uint16_t nc = (uint16_t)std::min(n_contigs, 65535u);
    const bool all_contigs = n_contigs > 65535u;
    for (int64_t i = lo; i < hi; ++i) {
        bad |= (unsigned)(!all_contigs & (ct[i] >= nc)) | (unsigned)(rl[i] == 0) | (unsigned)(al[i] == 0);
        k += rl[i] != al[i];
    }
    for (int64_t i = lo; i < hi; ++i)
        bad |= (unsigned)((uint64_t)ro[i] + rl[i] > alleles_len) | (unsigned)((uint64_t)ao[i] + al[i] > alleles_len);
    for (int64_t i = lo; i < hi; ++i) bad |= (unsigned)(pos[i] < 1);
    for (int64_t i = std::max<int64_t>(lo, 1); i < hi; ++i)
        bad |= (unsigned)(ct[i] < ct[i - 1]) | (unsigned)((ct[i] == ct[i - 1]) & (pos[i] < pos[i - 1]));
    *n_indel += k;
    return bad;
}

// rows [lo, hi) of `v`: 0 when every row is acceptable (and *n_indel += the rows whose alleles differ in length); otherwise
// the kind of the FIRST offending row (1 contig index, 2 empty allele, 3 allele outside the pool, 4 POS < 1, 5 order) and
// *row = its index.  Row lo is compared with row lo - 1 when lo > 0.
int validate_rows(const ugvc_variants* v, int64_t lo, int64_t hi, int n_contigs, int64_t* n_indel, int64_t* row) {
    constexpr int64_t kBlock = 8192;
    for (int64_t a = lo; a < hi; a += kBlock) {
        const int64_t b = std::min(a + kBlock, hi);
        int64_t k = 0;
        if (!block_bad(v->contig, v->pos, v->ref_len, v->alt_len, v->ref_off, v->alt_off, a, b, (unsigned)std::max(n_contigs, 0),
                       (uint64_t)v->alleles_len, &k)) {
            *n_indel += k;
            continue;
        }
        for (int64_t i = a; i < b; ++i) {
            int what = 0;
            if (v->contig[i] >= n_contigs) what = 1;
            else if (v->ref_len[i] == 0 || v->alt_len[i] == 0) what = 2;
            else if ((int64_t)v->ref_off[i] + v->ref_len[i] > v->alleles_len || (int64_t)v->alt_off[i] + v->alt_len[i] > v->alleles_len) what = 3;
            else if (v->pos[i] < 1) what = 4;
            else if (i && (v->contig[i] < v->contig[i - 1] || (v->contig[i] == v->contig[i - 1] && v->pos[i] < v->pos[i - 1]))) what = 5;
            if (what) {
                *row = i;
                return what;
            }
        }
    }
    return 0;
}

// Is the allele pool laid out the way a VCF reader leaves it - every row's REF then ALT, row after row, no gaps?  Then
// ref_off / alt_off of rows [lo, hi) follow from the lengths and the first row's offset, and the boundary call does not
// ship them (8 of a row's 39 bytes).  Row i is compared with row i - 1 for i > first (the chunk's first row: its offset is
// the chunk's base, handed to the device as a number).
__attribute__((target_clones("avx512f", "avx2", "default")))
static unsigned block_not_canonical(const uint16_t* __restrict__ rl, const uint16_t* __restrict__ al, const uint32_t* __restrict__ ro,
                                    const uint32_t* __restrict__ ao, int64_t lo, int64_t hi, int64_t first) {
    unsigned bad = 0;
    for (int64_t i = lo; i < hi; ++i) bad |= (unsigned)((uint64_t)ro[i] + rl[i] != (uint64_t)ao[i]);
    for (int64_t i = std::max(lo, first + 1); i < hi; ++i) bad |= (unsigned)((uint64_t)ao[i - 1] + al[i - 1] != (uint64_t)ro[i]);
    return bad;
}

bool offsets_canonical(const ugvc_variants* v, int64_t lo, int64_t hi, int64_t first) {
    return !block_not_canonical(v->ref_len, v->alt_len, v->ref_off, v->alt_off, lo, hi, first);
}

// memcpy whose destination is not read before it is written and not kept in the cache: a column piece goes from the caller's
// array into a pinned staging slot that only the DMA engine reads next (streaming stores: two memory transfers per byte
// instead of three).  Pieces below 4 KB and CPUs without AVX2 take memcpy.
__attribute__((target("avx2"))) static void copy_stream_avx2(uint8_t* dst, const uint8_t* src, size_t n) {
    const size_t head = (size_t)(-(intptr_t)dst & 63);
    memcpy(dst, src, head);
    dst += head; src += head; n -= head;
    const size_t body = n & ~(size_t)127;
    for (size_t i = 0; i < body; i += 128) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i + 32));
        const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i + 64));
        const __m256i d = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i + 96));
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i), a);
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 32), b);
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 64), c);
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 96), d);
    }
    memcpy(dst + body, src + body, n - body);
}

void copy_stream(void* dst, const void* src, size_t n) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (!avx2 || n < 4096) {
        memcpy(dst, src, n);
        return;
    }
    copy_stream_avx2(static_cast<uint8_t*>(dst), static_cast<const uint8_t*>(src), n);
}

void copy_stream_fence() { _mm_sfence(); }

const char* row_error_text(int what) {
    switch (what) {
        case 1: return "contig index out of range at row ";
        case 2: return "empty allele at row ";
        case 3: return "allele offset outside the pool at row ";
        case 4: return "POS must be >= 1 at row ";
        default: return "variants must be sorted by (contig, pos); row ";
    }
}

}  // namespace ugvc
