// Native VCF <-> SoA codec (include/ugvc_vcf.h).  Host threads only: BGZF blocks are independent deflate
// streams, so inflate / deflate parallelise per block; records are independent lines, so tokenising and
// the FILTER / INFO splice parallelise per line range.  Semantics mirror variantcalling_amd/io/vcf.py (the
// pure-Python host reference; tests/test_vcf_native.py compares both byte for byte).
#include "../../include/ugvc_vcf.h"

#include <zlib.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <numeric>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

thread_local std::string g_err;
int fail(std::string m) {
    g_err = std::move(m);
    return -1;
}

// Buffers of plain bytes that are written in full before they are read: resize() must not zero-fill them first (a 900 MB
// text buffer costs 0.1 s of serial memset before the parallel inflate starts)
template <class T>
struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    template <class U, class... A>
    void construct(U* p, A&&... a) {
        if constexpr (sizeof...(A) == 0) ::new (static_cast<void*>(p)) U;
        else ::new (static_cast<void*>(p)) U(std::forward<A>(a)...);
    }
};
using TextBuf = std::vector<char, NoInitAlloc<char>>;
using RawBuf = std::vector<unsigned char, NoInitAlloc<unsigned char>>;

// UGVC_VCF_TRACE=1: seconds per stage of the reader / writer on stderr
struct StageTimer {
    bool on = getenv("UGVC_VCF_TRACE") != nullptr;
    const char* who;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    explicit StageTimer(const char* w) : who(w) {}
    void lap(const char* what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[vcf] %s %-28s %.4f s\n", who, what, std::chrono::duration<double>(now - t).count());
        t = now;
    }
};

int pick_threads(int n_threads) {
    if (n_threads > 0) return std::min(n_threads, 256);
    const unsigned hw = std::thread::hardware_concurrency();
    // (64 at most by default: a tool runs several readers side by side; UGVC_VCF_MAX_THREADS is the measurement knob behind that
    // number - profiles/r06_reader_threads.txt)
    static const unsigned cap = [] { const char* e = getenv("UGVC_VCF_MAX_THREADS"); const int v = e ? atoi(e) : 0; return (unsigned)(v > 0 ? std::min(v, 256) : 64); }();
    return (int)std::min<unsigned>(hw ? hw : 1, cap);
}

// f(part, lo, hi) over [0, n) cut into `parts` contiguous ranges
template <class F>
void parallel_ranges(int64_t n, int parts, F f) {
    if (parts <= 1 || n < 2) {
        f(0, (int64_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    th.reserve(parts);
    for (int p = 0; p < parts; ++p) {
        const int64_t lo = n * p / parts, hi = n * (p + 1) / parts;
        th.emplace_back([=, &f] { f(p, lo, hi); });
    }
    for (auto& t : th) t.join();
}

// dynamic work queue: f(item) for item in [0, n)
template <class F>
void parallel_items(int64_t n, int threads, F f) {
    if (threads <= 1 || n < 2) {
        for (int64_t i = 0; i < n; ++i) f(i);
        return;
    }
    std::atomic<int64_t> next{0};
    std::vector<std::thread> th;
    const int T = (int)std::min<int64_t>(threads, n);
    for (int p = 0; p < T; ++p)
        th.emplace_back([&] {
            for (;;) {
                const int64_t i = next.fetch_add(1);
                if (i >= n) break;
                f(i);
            }
        });
    for (auto& t : th) t.join();
}

inline uint32_t le16(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
inline uint32_t le32(const unsigned char* p) { return le16(p) | (le16(p + 2) << 16); }

struct Block {
    size_t c_off, c_len;      // raw deflate payload
    size_t out_off;
    uint32_t out_len, crc;
};

// BGZF = gzip members with FEXTRA subfield 'B','C' (SLEN 2) = total member size - 1
bool parse_bgzf(const RawBuf& raw, std::vector<Block>& blocks) {
    size_t p = 0, out = 0;
    const size_t n = raw.size();
    while (p < n) {
        if (n - p < 18) return false;
        const unsigned char* h = raw.data() + p;
        if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return false;
        if (h[3] & ~4) return false;                       // FNAME / FCOMMENT / FHCRC: not BGZF-shaped
        const size_t xlen = le16(h + 10);
        if (n - p < 12 + xlen + 8) return false;
        size_t q = 12, bsize = 0;
        bool found = false;
        while (q + 4 <= 12 + xlen) {
            const size_t slen = le16(h + q + 2);
            if (h[q] == 'B' && h[q + 1] == 'C' && slen == 2 && q + 6 <= 12 + xlen) {
                bsize = le16(h + q + 4);
                found = true;
            }
            q += 4 + slen;
        }
        if (!found) return false;
        const size_t total = bsize + 1;
        if (total < 12 + xlen + 8 || n - p < total) return false;
        Block b;
        b.c_off = p + 12 + xlen;
        b.c_len = total - (12 + xlen) - 8;
        b.crc = le32(h + total - 8);
        b.out_len = le32(h + total - 4);
        b.out_off = out;
        out += b.out_len;
        blocks.push_back(b);
        p += total;
    }
    return true;
}

// ---- libdeflate, when the host has it (round 4) ------------------------------------------------------------------------
// BGZF blocks are whole, independent deflate streams of <= 64 KB: exactly what libdeflate's one-shot API is written for, at
// 2-3 x zlib's speed for the same ratio (the write-back of a 5 M-record callset was 0.57 s of zlib deflate out of 0.79 s).  The
// library is on the image without its header, so the five entry points are declared here and looked up with dlopen;
// zlib remains the fallback and the REFERENCE: `UGVC_DEFLATE=zlib` forces it, and only then are the compressed bytes those of
// the pure-Python codec (io/vcf.py drives zlib level 6 over the same 65280-byte blocks).  The inflated text, the CRCs, the
// block structure and therefore the tabix index's meaning are the same either way.
struct LibDeflate {
    void* (*alloc_c)(int) = nullptr;
    size_t (*compress)(void*, const void*, size_t, void*, size_t) = nullptr;
    size_t (*bound)(void*, size_t) = nullptr;
    void (*free_c)(void*) = nullptr;
    void* (*alloc_d)() = nullptr;
    int (*decompress)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
    void (*free_d)(void*) = nullptr;
    uint32_t (*crc)(uint32_t, const void*, size_t) = nullptr;
    bool ok = false;
};
std::atomic<int> g_deflate_backend{0};            // ugvc_vcf_set_deflate: 0 automatic, 1 zlib, 2 libdeflate
const LibDeflate& libdeflate_loaded() {
    static const LibDeflate L = [] {
        LibDeflate l;
        void* h = nullptr;
        for (const char* name : {"libdeflate.so.0", "libdeflate.so"})
            if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
        if (!h) return l;
        l.alloc_c = reinterpret_cast<void* (*)(int)>(dlsym(h, "libdeflate_alloc_compressor"));
        l.compress = reinterpret_cast<size_t (*)(void*, const void*, size_t, void*, size_t)>(dlsym(h, "libdeflate_deflate_compress"));
        l.bound = reinterpret_cast<size_t (*)(void*, size_t)>(dlsym(h, "libdeflate_deflate_compress_bound"));
        l.free_c = reinterpret_cast<void (*)(void*)>(dlsym(h, "libdeflate_free_compressor"));
        l.alloc_d = reinterpret_cast<void* (*)()>(dlsym(h, "libdeflate_alloc_decompressor"));
        l.decompress = reinterpret_cast<int (*)(void*, const void*, size_t, void*, size_t, size_t*)>(dlsym(h, "libdeflate_deflate_decompress"));
        l.free_d = reinterpret_cast<void (*)(void*)>(dlsym(h, "libdeflate_free_decompressor"));
        l.crc = reinterpret_cast<uint32_t (*)(uint32_t, const void*, size_t)>(dlsym(h, "libdeflate_crc32"));
        l.ok = l.alloc_c && l.compress && l.bound && l.free_c && l.alloc_d && l.decompress && l.free_d && l.crc;
        return l;
    }();
    return L;
}
const LibDeflate& libdeflate() {
    static const LibDeflate none;
    static const bool env_zlib = [] { const char* e = getenv("UGVC_DEFLATE"); return e && strcmp(e, "zlib") == 0; }();
    const int b = g_deflate_backend.load(std::memory_order_relaxed);
    if (b == 1 || (b == 0 && env_zlib)) return none;
    return libdeflate_loaded();
}

int inflate_serial(const RawBuf& raw, TextBuf& text, const std::string& path) {
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, 16 + MAX_WBITS) != Z_OK) return fail("zlib inflateInit2 failed");
    text.resize(std::max<size_t>(raw.size() * 4, 1 << 16));
    zs.next_in = const_cast<unsigned char*>(raw.data());
    zs.avail_in = (uInt)std::min<size_t>(raw.size(), 1u << 30);
    size_t in_done = 0, out_done = 0;
    for (;;) {
        if (out_done == text.size()) text.resize(text.size() * 2);
        const size_t room = std::min<size_t>(text.size() - out_done, 1u << 30);
        zs.next_out = reinterpret_cast<unsigned char*>(text.data() + out_done);
        zs.avail_out = (uInt)room;
        const size_t in_before = zs.avail_in;
        const int rc = inflate(&zs, Z_NO_FLUSH);
        in_done += in_before - zs.avail_in;
        out_done += room - zs.avail_out;
        if (zs.avail_in == 0 && in_done < raw.size()) {
            zs.next_in = const_cast<unsigned char*>(raw.data() + in_done);
            zs.avail_in = (uInt)std::min<size_t>(raw.size() - in_done, 1u << 30);
        }
        if (rc == Z_STREAM_END) {
            if (in_done >= raw.size()) break;
            if (inflateReset(&zs) != Z_OK) { inflateEnd(&zs); return fail(path + ": zlib inflateReset failed"); }
            continue;                                       // next gzip member
        }
        if (rc != Z_OK && rc != Z_BUF_ERROR) { inflateEnd(&zs); return fail(path + ": corrupt gzip stream"); }
        if (rc == Z_BUF_ERROR && zs.avail_in == 0 && in_done >= raw.size()) { inflateEnd(&zs); return fail(path + ": truncated gzip stream"); }
    }
    inflateEnd(&zs);
    text.resize(out_done);
    return 0;
}

int inflate_bgzf(const RawBuf& raw, const std::vector<Block>& blocks, TextBuf& text,
                 int threads, const std::string& path) {
    const size_t total = blocks.empty() ? 0 : blocks.back().out_off + blocks.back().out_len;
    text.resize(total);
    std::atomic<int> bad{0};
    // one inflate state per thread, reset from block to block
    std::atomic<int64_t> next_blk{0};
    const int64_t nb = (int64_t)blocks.size();
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(threads, nb));
    const LibDeflate& ld = libdeflate();
    parallel_ranges(T, T, [&](int, int64_t, int64_t) {
        if (ld.ok) {
            void* d = ld.alloc_d();
            if (!d) { bad = 1; return; }
            for (;;) {
                const int64_t i = next_blk.fetch_add(1);
                if (i >= nb) break;
                const Block& b = blocks[(size_t)i];
                if (b.out_len == 0) continue;
                size_t got = 0;
                const int rc = ld.decompress(d, raw.data() + b.c_off, b.c_len, text.data() + b.out_off, b.out_len, &got);
                if (rc != 0 || got != b.out_len) { bad = 1; continue; }
                if (ld.crc(0, text.data() + b.out_off, b.out_len) != b.crc) bad = 1;
            }
            ld.free_d(d);
            return;
        }
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) { bad = 1; return; }
        for (;;) {
            const int64_t i = next_blk.fetch_add(1);
            if (i >= nb) break;
            const Block& b = blocks[(size_t)i];
            if (b.out_len == 0) continue;
            if (inflateReset(&zs) != Z_OK) { bad = 1; break; }
            zs.next_in = const_cast<unsigned char*>(raw.data() + b.c_off);
            zs.avail_in = (uInt)b.c_len;
            zs.next_out = reinterpret_cast<unsigned char*>(text.data() + b.out_off);
            zs.avail_out = b.out_len;
            const int rc = inflate(&zs, Z_FINISH);
            if (rc != Z_STREAM_END || zs.avail_out != 0) { bad = 1; continue; }
            if ((uint32_t)crc32(0L, reinterpret_cast<const unsigned char*>(text.data() + b.out_off), b.out_len) != b.crc) bad = 1;
        }
        inflateEnd(&zs);
    });
    if (bad) return fail(path + ": corrupt BGZF block");
    return 0;
}

// whole file into memory, inflated if it is gzip (BGZF members in parallel, anything else serially).  The file is read by the
// worker threads in pieces (pread): one fread of a 3 GB FASTA is a single-threaded copy out of the page cache, ~1.6 GB/s, and
// for a plain-text file it was followed by a second serial copy from the byte buffer into the text buffer (round 4).
template <class Buf>
int read_file_parallel(const char* path, int fd, size_t size, int threads, Buf& dst) {
    dst.resize(size);
    if (size == 0) return 0;
    const int parts = (int)std::max<int64_t>(1, std::min<int64_t>(threads, (int64_t)(size >> 22)));      // pieces of >= 4 MB
    std::atomic<int> bad{0};
    parallel_ranges((int64_t)size, parts, [&](int, int64_t lo, int64_t hi) {
        char* p = reinterpret_cast<char*>(dst.data());
        int64_t at = lo;
        while (at < hi) {
            const ssize_t got = pread(fd, p + at, (size_t)(hi - at), (off_t)at);
            if (got <= 0) { bad = 1; return; }
            at += got;
        }
    });
    if (bad) return fail(std::string(path) + ": short read");
    return 0;
}

int load_text(const char* path, int threads, TextBuf& text) {
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return fail(std::string(path) + ": cannot open");
    struct Close { int fd; ~Close() { (void)close(fd); } } closer{fd};
    struct stat sb;
    if (fstat(fd, &sb) != 0 || sb.st_size < 0) return fail(std::string(path) + ": cannot stat");
    const size_t size = (size_t)sb.st_size;
    unsigned char magic[2] = {0, 0};
    const bool gz = size >= 2 && pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
    if (!gz) return read_file_parallel(path, fd, size, threads, text);
    RawBuf raw;
    if (read_file_parallel(path, fd, size, threads, raw)) return -1;
    std::vector<Block> blocks;
    if (parse_bgzf(raw, blocks)) return inflate_bgzf(raw, blocks, text, threads, path);
    return inflate_serial(raw, text, path);
}

struct Span {
    int64_t off;
    int32_t len;
};

inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }

// Python float(bytes): optional surrounding whitespace, full consumption; hex floats are not accepted
bool parse_float(const char* s, int len, double& out) {
    while (len > 0 && is_space(*s)) { ++s; --len; }
    while (len > 0 && is_space(s[len - 1])) --len;
    if (len <= 0 || len > 63) return false;
    // Fast path for what a VCF's numbers look like - [sign] digits [. digits], at most 15 digits in all: the digits as an integer
    // (exact in a double) divided by an exact power of ten is the correctly rounded value, i.e. what strtod returns (Clinger's
    // fast path); strtod itself - six calls per record - was half of the tokeniser's time.  Anything else takes the general path.
    {
        static const double p10[16] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15};
        int i = 0;
        const bool neg = s[0] == '-';
        if (s[0] == '-' || s[0] == '+') i = 1;
        uint64_t m = 0;
        int nd = 0, frac = 0;
        bool dot = false, ok = i < len;
        for (; i < len && ok; ++i) {
            const char c = s[i];
            if (c >= '0' && c <= '9') { m = m * 10 + (uint64_t)(c - '0'); ++nd; frac += dot ? 1 : 0; }
            else if (c == '.' && !dot) dot = true;
            else ok = false;
        }
        if (ok && nd >= 1 && nd <= 15) {
            const double v = frac ? (double)m / p10[frac] : (double)m;
            out = neg ? -v : v;
            return true;
        }
    }
    char buf[64];
    int m = 0;
    for (int i = 0; i < len; ++i) {
        const char c = s[i];
        if (c == 'x' || c == 'X' || c == 'p' || c == 'P' || c == '(') return false;
        if (c == '_') {                     // PEP 515: one underscore between two digits is skipped, any other is an error
            if (i == 0 || i + 1 >= len || s[i - 1] < '0' || s[i - 1] > '9' || s[i + 1] < '0' || s[i + 1] > '9') return false;
            continue;
        }
        buf[m++] = c;
    }
    buf[m] = 0;
    char* end = nullptr;
    const double v = strtod(buf, &end);
    if (end != buf + m) return false;
    out = v;
    return true;
}
inline double fnum(const char* s, int len) {
    double v;
    return parse_float(s, len, v) ? v : 0.0;
}
inline int32_t to_i32(double v) {                           // int(float): truncation toward zero
    if (!(v == v)) return 0;
    if (v >= 2147483647.0) return 2147483647;
    if (v <= -2147483648.0) return (int32_t)-2147483647 - 1;
    return (int32_t)v;
}

bool parse_int(const char* s, int len, int64_t& out) {       // Python int(bytes), base 10
    while (len > 0 && is_space(*s)) { ++s; --len; }
    while (len > 0 && is_space(s[len - 1])) --len;
    if (len <= 0) return false;
    bool neg = false;
    if (*s == '+' || *s == '-') { neg = *s == '-'; ++s; --len; }
    if (len <= 0 || len > 18) return false;
    int64_t v = 0;
    for (int i = 0; i < len; ++i) {
        if (s[i] == '_' && i > 0 && i + 1 < len && s[i - 1] >= '0' && s[i - 1] <= '9' && s[i + 1] >= '0' && s[i + 1] <= '9') continue;
        if (s[i] < '0' || s[i] > '9') return false;
        v = v * 10 + (s[i] - '0');
    }
    out = neg ? -v : v;
    return true;
}

const char* const kNewHeader[5] = {
    "##FILTER=<ID=LOW_SCORE,Description=\"Low decision tree score\">",
    "##FILTER=<ID=HPOL_RUN,Description=\"Homopolymer run\">",
    "##FILTER=<ID=COHORT_FP,Description=\"Common false positive in the cohort (blacklist)\">",
    "##INFO=<ID=TREE_SCORE,Number=1,Type=Float,Description=\"Filtering score\">",
    "##INFO=<ID=HPOL_RUN,Number=0,Type=Flag,Description=\"In or close to homopolymer run\">",
};

int format_f32(float x, char* buf, int cap) {
    if (cap < 8) return -1;
    if (x == 0.0f) { const char* z = std::signbit(x) ? "-0.0" : "0.0"; const int n = (int)strlen(z); memcpy(buf, z, (size_t)n + 1); return n; }
    if (!(x == x)) { memcpy(buf, "nan", 4); return 3; }
    if (std::isinf(x)) {
        const char* s = x < 0 ? "-inf" : "inf";
        const int n = (int)strlen(s);
        memcpy(buf, s, (size_t)n + 1);
        return n;
    }
    // shortest round-trip digits (scientific), laid out positionally with zero fill - what
    // numpy.format_float_positional(x, unique=True, trim="0") prints
    char sci[48];
    auto r = std::to_chars(sci, sci + sizeof sci, x, std::chars_format::scientific);
    if (r.ec != std::errc()) return -1;
    const char* p = sci;
    const bool neg = *p == '-';
    if (neg) ++p;
    char digits[24];
    int nd = 0;
    for (; p < r.ptr && *p != 'e'; ++p)
        if (*p != '.') digits[nd++] = *p;
    int ex = 0;
    if (p < r.ptr && *p == 'e') {
        ++p;
        const bool eneg = *p == '-';
        if (*p == '-' || *p == '+') ++p;
        for (; p < r.ptr; ++p) ex = ex * 10 + (*p - '0');
        if (eneg) ex = -ex;
    }
    while (nd > 1 && digits[nd - 1] == '0') --nd;
    const int need = (neg ? 1 : 0) + (ex >= 0 ? std::max(nd, ex + 1) + 2 : -ex + nd + 2) + 1;
    if (need > cap) return -1;
    int n = 0;
    if (neg) buf[n++] = '-';
    if (ex >= 0) {
        for (int i = 0; i <= ex; ++i) buf[n++] = i < nd ? digits[i] : '0';
        buf[n++] = '.';
        if (nd > ex + 1) for (int i = ex + 1; i < nd; ++i) buf[n++] = digits[i];
        else buf[n++] = '0';
    } else {
        buf[n++] = '0';
        buf[n++] = '.';
        for (int i = 0; i < -ex - 1; ++i) buf[n++] = '0';
        for (int i = 0; i < nd; ++i) buf[n++] = digits[i];
    }
    buf[n] = 0;
    return n;
}

}  // namespace

// Column vectors are sized once and then written in full by the worker threads: NoInitAlloc (above) leaves their elements
// uninitialised (a `resize` of twenty 5 M-element columns was a serial zero fill of ~150 MB - and the first touch of every
// page by ONE thread; now the pages are touched by the threads that fill them).
template <class T> using Col = std::vector<T, NoInitAlloc<T>>;

struct ugvc_vcf {
    std::string path;
    TextBuf text;
    std::vector<Span> hdr_lines;            // file order
    Col<Span> rec_lines;                    // file order, without trailing \r / \n
    std::string header_joined;
    int64_t n = 0;
    int64_t n_total = 0, part_lo = 0;        // ugvc_vcf_read_part: records in the file, first record (file order) of this part
    // table order
    Col<uint16_t> contig;
    Col<uint8_t> gq, gt, has_id, alleles, n_alt;
    Col<int32_t> pos, dp, ad_ref, ad_alt, filter_len, rec_len;
    Col<uint16_t> ref_len, alt_len;
    Col<uint32_t> ref_off, alt_off;
    Col<float> qual, sor, tlod;
    Col<int64_t> order, filter_off, rec_off;
    std::vector<std::string> contig_names;  // index = contig column (for the tabix index of the output)
};

namespace {

struct Parsed {                              // file order
    Col<uint16_t> contig;
    Col<uint8_t> gq, gt, has_id, n_alt;
    Col<int32_t> pos, dp, adr, ada;
    Col<float> qual, sor, tlod;
    Col<Span> ref, alt, filt;
    void resize(size_t n) {
        contig.resize(n); gq.resize(n); gt.resize(n); has_id.resize(n); n_alt.resize(n);
        pos.resize(n); dp.resize(n); adr.resize(n); ada.resize(n);
        qual.resize(n); sor.resize(n); tlod.resize(n);
        ref.resize(n); alt.resize(n); filt.resize(n);
    }
};

// one record line -> file-order columns; returns false with a message on malformed input
struct ContigMemo { const char* name = nullptr; int len = 0; int idx = 0; };     // the previous record's CHROM (files are sorted: no hash per record)

bool parse_record(const char* base, Span line, int64_t k, const std::unordered_map<std::string_view, int>& contig_idx,
                  int sample, Parsed& P, const std::string& path, std::string& err, ContigMemo& memo) {
    const char* s = base + line.off;
    const char* e = s + line.len;
    const int want = 10 + sample;
    Span f[16];
    int nf = 0;
    Span fs{0, 0};                           // the sample column (field 9 + sample)
    {
        const char* p = s;
        for (;;) {
            const char* t = static_cast<const char*>(memchr(p, '\t', (size_t)(e - p)));
            const char* fe = t ? t : e;
            if (nf < 9) f[nf] = Span{(int64_t)(p - base), (int32_t)(fe - p)};
            if (nf == want - 1) fs = Span{(int64_t)(p - base), (int32_t)(fe - p)};
            ++nf;
            if (!t) break;
            p = t + 1;
        }
    }
    if (nf < 8) {
        err = path + ": record " + std::to_string(k + 1) + " has " + std::to_string(nf) + " columns";
        return false;
    }
    if (!(memo.name && memo.len == f[0].len && memcmp(memo.name, base + f[0].off, (size_t)f[0].len) == 0)) {
        auto it = contig_idx.find(std::string_view(base + f[0].off, (size_t)f[0].len));
        if (it == contig_idx.end()) {
            err = path + ": contig '" + std::string(base + f[0].off, (size_t)f[0].len) + "' is not in the reference";
            return false;
        }
        memo.name = base + f[0].off; memo.len = f[0].len; memo.idx = it->second;
    }
    P.contig[k] = (uint16_t)memo.idx;
    int64_t pv;
    if (!parse_int(base + f[1].off, f[1].len, pv) || pv < INT32_MIN || pv > INT32_MAX) {
        err = path + ": record " + std::to_string(k + 1) + ": POS '" + std::string(base + f[1].off, (size_t)f[1].len) + "' is not an integer";
        return false;
    }
    P.pos[k] = (int32_t)pv;
    P.has_id[k] = !(f[2].len == 1 && base[f[2].off] == '.');
    P.ref[k] = f[3];
    {
        const char* a = base + f[4].off;
        const char* c = static_cast<const char*>(memchr(a, ',', (size_t)f[4].len));
        P.alt[k] = Span{f[4].off, c ? (int32_t)(c - a) : f[4].len};
        int na = 1;                                      // ALT alleles of the record (multi-allelic rows are expanded by the host)
        for (int q = 0; q < f[4].len; ++q) na += a[q] == ',';
        P.n_alt[k] = (uint8_t)(na > 255 ? 255 : na);
    }
    if (P.ref[k].len == 0 || P.alt[k].len == 0) {           // (schema.VariantTable.validate's rule, enforced where the row is known)
        err = path + ": record " + std::to_string(k + 1) + ": empty alleles are not representable";
        return false;
    }
    if (P.ref[k].len > 65535 || P.alt[k].len > 65535) {
        err = path + ": record " + std::to_string(k + 1) + ": allele longer than 65535 bases";
        return false;
    }
    P.qual[k] = (float)fnum(base + f[5].off, f[5].len);
    P.filt[k] = f[6];
    float sor = 0.f, tlod = 0.f;
    {
        const char* p = base + f[7].off;
        const char* ie = p + f[7].len;
        while (p <= ie) {
            const char* t = static_cast<const char*>(memchr(p, ';', (size_t)(ie - p)));
            const char* fe = t ? t : ie;
            const int len = (int)(fe - p);
            if (len >= 4 && memcmp(p, "SOR=", 4) == 0) sor = (float)fnum(p + 4, len - 4);
            else if (len >= 5 && memcmp(p, "TLOD=", 5) == 0) {
                double best = 0.0;
                bool first = true;
                const char* q = p + 5;
                while (q <= fe) {
                    const char* c = static_cast<const char*>(memchr(q, ',', (size_t)(fe - q)));
                    const char* ce = c ? c : fe;
                    const double v = fnum(q, (int)(ce - q));
                    if (first || v > best) best = v;       // Python max(): keeps the first of equals, NaN never wins later
                    first = false;
                    if (!c) break;
                    q = c + 1;
                }
                tlod = (float)best;
            }
            if (!t) break;
            p = t + 1;
        }
    }
    P.sor[k] = sor;
    P.tlod[k] = tlod;
    int32_t dp = 0, adr = 0, ada = 0;
    int gq = 0, gt = 0;
    if (nf > 9 + sample && nf > 9) {
        const char* kp = base + f[8].off;
        const char* ke = kp + f[8].len;
        const char* vp = base + fs.off;
        const char* ve = vp + fs.len;
        bool kdone = false, vdone = false;
        while (!kdone && !vdone) {                          // zip(keys, vals)
            const char* kt = static_cast<const char*>(memchr(kp, ':', (size_t)(ke - kp)));
            const char* kfe = kt ? kt : ke;
            const char* vt = static_cast<const char*>(memchr(vp, ':', (size_t)(ve - vp)));
            const char* vfe = vt ? vt : ve;
            const int kl = (int)(kfe - kp), vl = (int)(vfe - vp);
            if (kl == 2 && kp[0] == 'D' && kp[1] == 'P') dp = to_i32(fnum(vp, vl));
            else if (kl == 2 && kp[0] == 'A' && kp[1] == 'D') {
                const char* c = static_cast<const char*>(memchr(vp, ',', (size_t)vl));
                if (!c) { adr = to_i32(fnum(vp, vl)); ada = 0; }
                else {
                    adr = to_i32(fnum(vp, (int)(c - vp)));
                    const char* c2 = static_cast<const char*>(memchr(c + 1, ',', (size_t)(vfe - c - 1)));
                    ada = to_i32(fnum(c + 1, (int)((c2 ? c2 : vfe) - c - 1)));
                }
            } else if (kl == 2 && kp[0] == 'G' && kp[1] == 'Q') {
                const int32_t g = to_i32(fnum(vp, vl));
                gq = g < 0 ? 0 : (g > 255 ? 255 : g);
            } else if (kl == 2 && kp[0] == 'G' && kp[1] == 'T') {
                // val.replace("|", "/").split("/"): 2 iff exactly ["1", "1"], else 1 iff any allele == "1"
                int n_all = 0, n_one = 0;
                const char* q = vp;
                for (;;) {
                    const char* c = q;
                    while (c < vfe && *c != '/' && *c != '|') ++c;
                    ++n_all;
                    if (c - q == 1 && *q == '1') ++n_one;
                    if (c >= vfe) break;
                    q = c + 1;
                }
                gt = (n_all == 2 && n_one == 2) ? 2 : (n_one > 0 ? 1 : 0);
            }
            if (!kt) kdone = true; else kp = kt + 1;
            if (!vt) vdone = true; else vp = vt + 1;
        }
    }
    P.dp[k] = dp; P.adr[k] = adr; P.ada[k] = ada;
    P.gq[k] = (uint8_t)gq; P.gt[k] = (uint8_t)gt;
    return true;
}

}  // namespace

namespace {

// ---- tabix index of the written file (the reference ends its write loop with pysam.tabix_index:
// ugvc/pipelines/vcfbed/calibrate_bridging_snvs.py:130).  Format: SAMv1 / tabix specification - binning index
// (min shift 14, depth 5) + 16 kb linear index over BGZF virtual offsets, format VCF (columns 1, 2; END= honoured),
// the whole index BGZF-compressed.  Skipped (with a 1 return) when the records are not grouped by contig and sorted
// by position, which tabix itself refuses.
int reg2bin(int64_t beg, int64_t end) {
    --end;
    if (beg >> 14 == end >> 14) return ((1 << 15) - 1) / 7 + (int)(beg >> 14);
    if (beg >> 17 == end >> 17) return ((1 << 12) - 1) / 7 + (int)(beg >> 17);
    if (beg >> 20 == end >> 20) return ((1 << 9) - 1) / 7 + (int)(beg >> 20);
    if (beg >> 23 == end >> 23) return ((1 << 6) - 1) / 7 + (int)(beg >> 23);
    if (beg >> 26 == end >> 26) return ((1 << 3) - 1) / 7 + (int)(beg >> 26);
    return 0;
}

void put32(std::string& o, uint32_t x) { for (int i = 0; i < 4; ++i) o.push_back((char)(x >> (8 * i))); }
void put64(std::string& o, uint64_t x) { for (int i = 0; i < 8; ++i) o.push_back((char)(x >> (8 * i))); }

int bgzf_write_all(const std::string& data, const std::string& path) {
    FILE* fh = fopen(path.c_str(), "wb");
    if (!fh) return fail(path + ": cannot open for writing");
    bool ok = true;
    constexpr size_t kBlk = 65280;
    for (size_t lo = 0; lo < data.size() && ok; lo += kBlk) {
        const size_t len = std::min(kBlk, data.size() - lo);
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { ok = false; break; }
        std::string o(18 + deflateBound(&zs, (uLong)len) + 8, '\0');
        zs.next_in = reinterpret_cast<unsigned char*>(const_cast<char*>(data.data() + lo));
        zs.avail_in = (uInt)len;
        zs.next_out = reinterpret_cast<unsigned char*>(&o[18]);
        zs.avail_out = (uInt)(o.size() - 26);
        const int rc = deflate(&zs, Z_FINISH);
        const size_t clen = zs.total_out;
        deflateEnd(&zs);
        if (rc != Z_STREAM_END) { ok = false; break; }
        static const unsigned char hd[16] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 'B', 'C', 0x02, 0};
        memcpy(&o[0], hd, 16);
        const uint32_t bsize = (uint32_t)(clen + 25);
        o[16] = (char)(bsize & 0xff); o[17] = (char)(bsize >> 8);
        const uint32_t crc = (uint32_t)crc32(0L, reinterpret_cast<const unsigned char*>(data.data() + lo), (uInt)len);
        unsigned char* t = reinterpret_cast<unsigned char*>(&o[18 + clen]);
        for (int i = 0; i < 4; ++i) { t[i] = (unsigned char)(crc >> (8 * i)); t[4 + i] = (unsigned char)((uint32_t)len >> (8 * i)); }
        o.resize(18 + clen + 8);
        ok = fwrite(o.data(), 1, o.size(), fh) == o.size();
    }
    static const unsigned char kEof[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0,
                                           0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (ok) ok = fwrite(kEof, 1, 28, fh) == 28;
    if (fclose(fh) != 0) ok = false;
    return ok ? 0 : fail(path + ": write failed");
}

// INFO/END of a record (symbolic alleles): "END=" at the start of INFO [q, ie) or after a ';'; -1 when absent
inline int64_t info_end_of(const char* q, const char* ie) {
    for (const char* p = q; p + 4 <= ie;) {
        if (memcmp(p, "END=", 4) == 0 && (p == q || p[-1] == ';')) {
            int64_t v = 0;
            const char* d = p + 4;
            bool any = false;
            while (d < ie && *d >= '0' && *d <= '9') { v = v * 10 + (*d - '0'); ++d; any = true; }
            return any ? v : -1;
        }
        const char* nx = static_cast<const char*>(memchr(p, ';', (size_t)(ie - p)));
        if (!nx) break;
        p = nx + 1;
    }
    return -1;
}

// `info_end`: INFO/END per record in file order (-1: none), found by the writer's formatting threads while they had the
// INFO field at hand - the index pass itself is one light serial loop (it re-read every record's text before: 0.27 s of a
// 5 M-record write-back on one thread)
int write_tbi(const ugvc_vcf* h, const char* out_path, const std::vector<int64_t>& row_of, const std::vector<uint32_t>& out_len,
              const std::vector<int64_t>& info_end, size_t header_len, const std::vector<uint32_t>& blk_clen, int /*threads*/) {
    const int64_t n = h->n;
    constexpr uint64_t kBlk = 65280;
    std::vector<uint64_t> coff(blk_clen.size() + 1, 0);
    for (size_t b = 0; b < blk_clen.size(); ++b) coff[b + 1] = coff[b] + blk_clen[b];
    auto voff = [&](uint64_t u) { return (coff[(size_t)(u / kBlk)] << 16) | (u % kBlk); };
    struct Ref {
        std::vector<std::pair<uint32_t, std::pair<uint64_t, uint64_t>>> chunks;   // (bin, [beg, end)) in file order
        std::vector<uint64_t> lin;
    };
    std::vector<Ref> refs;
    std::vector<int> tid_of(h->contig_names.size(), -1);
    std::vector<int> name_of_tid;
    int cur_c = -1;
    int64_t last_beg = -1;
    uint64_t u = header_len;
    for (int64_t j = 0; j < n; ++j) {
        const size_t k = (size_t)row_of[(size_t)j];
        const int c = h->contig[k];
        const int64_t beg = (int64_t)h->pos[k] - 1;
        int64_t end = beg + h->ref_len[k];
        if (info_end[(size_t)j] > beg) end = info_end[(size_t)j];
        if (beg < 0) return 1;
        if (c != cur_c) {
            if (tid_of[(size_t)c] >= 0) return 1;                 // contig seen before: not grouped, tabix would refuse
            tid_of[(size_t)c] = (int)refs.size();
            name_of_tid.push_back(c);
            refs.emplace_back();
            cur_c = c;
            last_beg = -1;
        }
        if (beg < last_beg) return 1;                             // unsorted
        last_beg = beg;
        Ref& r = refs.back();
        const uint64_t v0 = voff(u), v1 = voff(u + out_len[(size_t)j]);
        const uint32_t bin = (uint32_t)reg2bin(beg, end > beg ? end : beg + 1);
        if (!r.chunks.empty() && r.chunks.back().first == bin) r.chunks.back().second.second = v1;
        else r.chunks.push_back({bin, {v0, v1}});
        const size_t w0 = (size_t)(beg >> 14), w1 = (size_t)(((end > beg ? end : beg + 1) - 1) >> 14);
        if (r.lin.size() <= w1) r.lin.resize(w1 + 1, ~0ull);
        for (size_t w = w0; w <= w1; ++w)
            if (r.lin[w] == ~0ull) r.lin[w] = v0;
        u += out_len[(size_t)j];
    }
    std::string o;
    o.append("TBI\1", 4);
    put32(o, (uint32_t)refs.size());
    put32(o, 2); put32(o, 1); put32(o, 2); put32(o, 0); put32(o, '#'); put32(o, 0);
    std::string nm;
    for (int c : name_of_tid) { nm.append(h->contig_names[(size_t)c]); nm.push_back('\0'); }
    put32(o, (uint32_t)nm.size());
    o.append(nm);
    for (Ref& r : refs) {
        // group the chunks by bin (a bin's chunks stay in file order)
        std::vector<size_t> idx(r.chunks.size());
        std::iota(idx.begin(), idx.end(), (size_t)0);
        std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return r.chunks[a].first < r.chunks[b].first; });
        uint32_t n_bin = 0;
        for (size_t i = 0; i < idx.size(); ++i) n_bin += i == 0 || r.chunks[idx[i]].first != r.chunks[idx[i - 1]].first;
        put32(o, n_bin);
        for (size_t i = 0; i < idx.size();) {
            size_t e = i;
            while (e < idx.size() && r.chunks[idx[e]].first == r.chunks[idx[i]].first) ++e;
            put32(o, r.chunks[idx[i]].first);
            put32(o, (uint32_t)(e - i));
            for (size_t q = i; q < e; ++q) { put64(o, r.chunks[idx[q]].second.first); put64(o, r.chunks[idx[q]].second.second); }
            i = e;
        }
        for (size_t w = 1; w < r.lin.size(); ++w) if (r.lin[w] == ~0ull) r.lin[w] = r.lin[w - 1];
        if (!r.lin.empty() && r.lin[0] == ~0ull) r.lin[0] = 0;
        for (size_t w = 1; w < r.lin.size(); ++w) if (r.lin[w] == ~0ull) r.lin[w] = r.lin[w - 1];
        put32(o, (uint32_t)r.lin.size());
        for (uint64_t x : r.lin) put64(o, x);
    }
    put64(o, 0);                                                   // records without coordinates
    return bgzf_write_all(o, std::string(out_path) + ".tbi");
}

}  // namespace

extern "C" {

const char* ugvc_vcf_last_error(void) { return g_err.c_str(); }
int ugvc_vcf_abi_version(void) { return 2; }   // 2: contig column u16
int ugvc_vcf_set_deflate(int backend) {
    if (backend < 0 || backend > 2) return fail("ugvc_vcf_set_deflate: backend must be 0 (automatic), 1 (zlib) or 2 (libdeflate)");
    if (backend == 2 && !libdeflate_loaded().ok) return fail("ugvc_vcf_set_deflate: libdeflate.so.0 is not on this host");
    g_deflate_backend.store(backend);
    return libdeflate().ok ? 2 : 1;
}
int ugvc_vcf_format_f32(float x, char* buf, int cap) { return buf ? format_f32(x, buf, cap) : -1; }

// ---- the codec's own file reader, bare (round 6): plain / gzip / BGZF file -> its bytes ----------------------------
// What every reader above starts with (load_text: BGZF members inflated one block per thread through the selected deflate back
// end), exposed so that it can be checked on material the codec did not write itself: tests/test_htslib_bgzf.py inflates a
// real htslib stream (the reference's own `.vcf.gz.csi`) with both back ends and compares with Python's gzip.
struct ugvc_blob { TextBuf bytes; };
int ugvc_bgzf_read(const char* path, int n_threads, ugvc_blob** out, const char** data, int64_t* len) {
    if (!path || !out || !data || !len) return fail("NULL argument");
    *out = nullptr;
    try {
        std::unique_ptr<ugvc_blob> b(new ugvc_blob());
        if (load_text(path, pick_threads(n_threads), b->bytes)) return -1;
        *data = b->bytes.data();
        *len = (int64_t)b->bytes.size();
        *out = b.release();
        return 0;
    } catch (const std::exception& e) {
        return fail(std::string(path) + ": " + e.what());
    }
}
void ugvc_blob_free(ugvc_blob* b) { delete b; }

void ugvc_vcf_free(ugvc_vcf* h) { delete h; }

// ---- "the reader knows how many records it holds" hook (round 4) --------------------------------------------------
// A tool that wants to prepare something of the callset's size while the reader is still tokenising (the GPU engine's
// ugvc_reserve) leaves a function here: it is called ONCE, on the calling thread, by the next ugvc_vcf_read / _read_part of
// THIS thread as soon as the record lines are counted - with two thirds of the reader's time still ahead - and forgotten.
namespace {
thread_local void (*t_count_hook)(int64_t, int64_t, void*) = nullptr;
thread_local void* t_count_user = nullptr;
}
void ugvc_vcf_set_count_hook(void (*fn)(int64_t n_records, int64_t text_bytes, void* user), void* user) {
    t_count_hook = fn;
    t_count_user = user;
}

int ugvc_vcf_read(const char* path, const char* const* contig_names, int n_contigs, int is_mutect, int sample,
                  int n_threads, ugvc_vcf** out) {
    return ugvc_vcf_read_part(path, contig_names, n_contigs, is_mutect, sample, n_threads, 0, 1, out);
}

int ugvc_vcf_read_part(const char* path, const char* const* contig_names, int n_contigs, int is_mutect, int sample,
                       int n_threads, int part, int n_parts, ugvc_vcf** out) {
    if (!path || !out || (n_contigs > 0 && !contig_names)) return fail("NULL argument");
    if (n_parts < 1 || part < 0 || part >= n_parts) return fail("ugvc_vcf_read_part: need 0 <= part < n_parts");
    if (n_contigs < 0 || n_contigs > 65535) return fail("the contig column is u16: at most 65535 contigs");
    if (sample < 0 || sample > 1000000) return fail("bad sample index");
    *out = nullptr;
    const int threads = pick_threads(n_threads);
    std::unique_ptr<ugvc_vcf> h(new ugvc_vcf());
    h->path = path;
    StageTimer st("read");
    if (load_text(path, threads, h->text)) return -1;
    st.lap("load + inflate");
    const char* base = h->text.data();
    const int64_t tn = (int64_t)h->text.size();

    // ---- lines: part p owns the lines that START inside its byte range
    const int parts = (int)std::max<int64_t>(1, std::min<int64_t>(threads, tn / (1 << 16)));
    std::vector<std::vector<Span>> hdr_p((size_t)parts), rec_p((size_t)parts);
    std::vector<int64_t> too_long((size_t)parts, -1);                 // byte offset of a line longer than a Span can hold
    parallel_ranges(tn, parts, [&](int p, int64_t lo, int64_t hi) {
        int64_t s = lo;
        if (lo > 0) {
            const void* nl = memchr(base + lo - 1, '\n', (size_t)(tn - lo + 1));
            if (!nl) return;
            s = (const char*)nl - base + 1;
        }
        bool sized = false;
        while (s < hi && s < tn) {
            const void* nl = memchr(base + s, '\n', (size_t)(tn - s));
            const int64_t e = nl ? (const char*)nl - base : tn;
            if (!sized && base[s] != '#') {                       // (one allocation per part: lines of about this length to the part's end)
                rec_p[(size_t)p].reserve((size_t)((hi - s) / std::max<int64_t>(e - s + 1, 16) * 5 / 4 + 16));
                sized = true;
            }
            int64_t le = e;
            while (le > s && (base[le - 1] == '\r' || base[le - 1] == '\n')) --le;
            if (base[s] == '#' && e > s) hdr_p[(size_t)p].push_back(Span{s, (int32_t)(le - s)});
            else {
                bool blank = true;
                for (int64_t q = s; q < le && blank; ++q) blank = is_space(base[q]);
                if (!blank) {
                    if (le - s > INT32_MAX) { too_long[(size_t)p] = s; return; }
                    rec_p[(size_t)p].push_back(Span{s, (int32_t)(le - s)});
                }
            }
            s = e + 1;
        }
    });
    for (int p = 0; p < parts; ++p)
        if (too_long[(size_t)p] >= 0)                                 // (reported, not silently dropped with the rest of the part)
            return fail(h->path + ": line at byte " + std::to_string(too_long[(size_t)p]) + " is longer than 2 GiB");
    // (the parts' record lines go to their places in parallel: a serial append of 5 M spans was 80 MB of copying on one thread)
    std::vector<size_t> rec_at((size_t)parts + 1, 0);
    for (int p = 0; p < parts; ++p) {
        h->hdr_lines.insert(h->hdr_lines.end(), hdr_p[(size_t)p].begin(), hdr_p[(size_t)p].end());
        rec_at[(size_t)p + 1] = rec_at[(size_t)p] + rec_p[(size_t)p].size();
    }
    h->rec_lines.resize(rec_at[(size_t)parts]);
    parallel_ranges(parts, parts, [&](int, int64_t lo, int64_t hi) {
        for (int64_t p = lo; p < hi; ++p)
            if (!rec_p[(size_t)p].empty())
                memcpy(h->rec_lines.data() + rec_at[(size_t)p], rec_p[(size_t)p].data(), rec_p[(size_t)p].size() * sizeof(Span));
    });
    for (size_t i = 0; i < h->hdr_lines.size(); ++i) {
        if (i) h->header_joined.push_back('\n');
        h->header_joined.append(base + h->hdr_lines[i].off, (size_t)h->hdr_lines[i].len);
    }
    // a PART of the callset (one rank of a multi-process run): equal-count slices of the record lines in FILE order - the
    // ranks' parts are the equal-count shards of the sorted callset when the file is sorted, which the caller verifies across
    // ranks; everything below (tokeniser, order, columns) then works on 1 / n_parts of the records
    h->n_total = (int64_t)h->rec_lines.size();
    if (n_parts > 1) {
        const int64_t R = h->n_total, qb = R / n_parts, qr = R % n_parts;
        const int64_t lo = (int64_t)part * qb + std::min<int64_t>(part, qr), hi = lo + qb + (part < qr ? 1 : 0);
        Col<Span> mine(h->rec_lines.begin() + lo, h->rec_lines.begin() + hi);
        h->rec_lines.swap(mine);
        h->part_lo = lo;
    }
    const int64_t n = (int64_t)h->rec_lines.size();
    h->n = n;
    st.lap("lines");
    if (t_count_hook) {
        void (*fn)(int64_t, int64_t, void*) = t_count_hook;
        t_count_hook = nullptr;
        fn(n, tn, t_count_user);
    }

    // ---- tokenise (file order)
    std::unordered_map<std::string_view, int> contig_idx;
    std::vector<std::string> names((size_t)n_contigs);
    for (int c = 0; c < n_contigs; ++c) names[(size_t)c] = contig_names[c] ? contig_names[c] : "";
    for (int c = 0; c < n_contigs; ++c) contig_idx[std::string_view(names[(size_t)c])] = c;   // duplicates: the last wins
    h->contig_names = names;
    Parsed P;
    P.resize((size_t)n);
    const int pparts = (int)std::max<int64_t>(1, std::min<int64_t>(threads, n / 2048));
    std::vector<std::string> errs((size_t)pparts);
    std::vector<int64_t> err_at((size_t)pparts, INT64_MAX);
    parallel_ranges(n, pparts, [&](int p, int64_t lo, int64_t hi) {
        ContigMemo memo;
        for (int64_t k = lo; k < hi; ++k)
            if (!parse_record(base, h->rec_lines[(size_t)k], k, contig_idx, sample, P, h->path, errs[(size_t)p], memo)) {
                err_at[(size_t)p] = k;
                return;
            }
    });
    for (int p = 0; p < pparts; ++p)
        if (err_at[(size_t)p] != INT64_MAX) return fail(errs[(size_t)p]);     // parts are in file order: first error first
    if (is_mutect)
        for (int64_t k = 0; k < n; ++k) P.qual[(size_t)k] = 10.0f * P.tlod[(size_t)k];
    st.lap("tokenise");

    // ---- stable order by (contig, pos)
    h->order.resize((size_t)n);
    {
        auto key_of = [&](int64_t k) {
            return ((uint64_t)P.contig[(size_t)k] << 32) | (uint32_t)((uint32_t)P.pos[(size_t)k] ^ 0x80000000u);
        };
        // (a sorted file - the usual case - is recognised by the parts in parallel, seams included, without a key array)
        std::vector<char> part_sorted((size_t)pparts, 1);
        parallel_ranges(n, pparts, [&](int p, int64_t lo, int64_t hi) {
            uint64_t prev = lo > 0 ? key_of(lo - 1) : 0;
            bool ok = true;
            for (int64_t k = lo; k < hi; ++k) {
                h->order[(size_t)k] = k;
                const uint64_t x = key_of(k);
                ok &= x >= prev;
                prev = x;
            }
            part_sorted[(size_t)p] = ok;
        });
        bool sorted = true;
        for (char c : part_sorted) sorted &= c != 0;
        if (!sorted) {
            Col<uint64_t> key((size_t)n);
            parallel_ranges(n, pparts, [&](int, int64_t lo, int64_t hi) { for (int64_t k = lo; k < hi; ++k) key[(size_t)k] = key_of(k); });
            std::stable_sort(h->order.begin(), h->order.end(), [&](int64_t a, int64_t b) { return key[(size_t)a] < key[(size_t)b]; });
        }
    }
    st.lap("order");

    // ---- table columns
    h->contig.resize((size_t)n); h->gq.resize((size_t)n); h->gt.resize((size_t)n); h->has_id.resize((size_t)n);
    h->n_alt.resize((size_t)n); h->rec_off.resize((size_t)n); h->rec_len.resize((size_t)n);
    h->pos.resize((size_t)n); h->dp.resize((size_t)n); h->ad_ref.resize((size_t)n); h->ad_alt.resize((size_t)n);
    h->ref_len.resize((size_t)n); h->alt_len.resize((size_t)n); h->ref_off.resize((size_t)n); h->alt_off.resize((size_t)n);
    h->qual.resize((size_t)n); h->sor.resize((size_t)n); h->tlod.resize((size_t)n);
    h->filter_off.resize((size_t)n); h->filter_len.resize((size_t)n);
    // allele-pool offsets in table order: per-part byte counts, a scan over the parts, the offsets of every part in parallel
    uint64_t tot = 0;
    {
        std::vector<uint64_t> part_bytes((size_t)pparts, 0);
        parallel_ranges(n, pparts, [&](int p, int64_t lo, int64_t hi) {
            uint64_t b = 0;
            for (int64_t k = lo; k < hi; ++k) {
                const size_t j = (size_t)h->order[(size_t)k];
                b += (uint64_t)P.ref[j].len + (uint64_t)P.alt[j].len;
            }
            part_bytes[(size_t)p] = b;
        });
        for (int p = 0; p < pparts; ++p) { const uint64_t b = part_bytes[(size_t)p]; part_bytes[(size_t)p] = tot; tot += b; }
        if (tot > 0xFFFFFFFFull) return fail(h->path + ": allele pool exceeds 4 GiB");
        parallel_ranges(n, pparts, [&](int p, int64_t lo, int64_t hi) {
            uint64_t t = part_bytes[(size_t)p];
            for (int64_t k = lo; k < hi; ++k) {
                const size_t j = (size_t)h->order[(size_t)k];
                h->ref_off[(size_t)k] = (uint32_t)t;
                h->alt_off[(size_t)k] = (uint32_t)(t + (uint64_t)P.ref[j].len);
                t += (uint64_t)P.ref[j].len + (uint64_t)P.alt[j].len;
            }
        });
    }
    h->alleles.resize((size_t)tot);
    st.lap("allocate + allele offsets");
    uint8_t code[256];
    memset(code, 0, sizeof code);
    code['A'] = code['a'] = 1; code['C'] = code['c'] = 2; code['G'] = code['g'] = 3; code['T'] = code['t'] = 4;
    parallel_ranges(n, pparts, [&](int, int64_t lo, int64_t hi) {
        for (int64_t k = lo; k < hi; ++k) {
            const size_t j = (size_t)h->order[(size_t)k], kk = (size_t)k;
            h->contig[kk] = P.contig[j]; h->pos[kk] = P.pos[j]; h->gq[kk] = P.gq[j]; h->gt[kk] = P.gt[j];
            h->n_alt[kk] = P.n_alt[j]; h->rec_off[kk] = h->rec_lines[j].off; h->rec_len[kk] = h->rec_lines[j].len;
            h->has_id[kk] = P.has_id[j]; h->dp[kk] = P.dp[j]; h->ad_ref[kk] = P.adr[j]; h->ad_alt[kk] = P.ada[j];
            h->qual[kk] = P.qual[j]; h->sor[kk] = P.sor[j]; h->tlod[kk] = P.tlod[j];
            h->ref_len[kk] = (uint16_t)P.ref[j].len; h->alt_len[kk] = (uint16_t)P.alt[j].len;
            h->filter_off[kk] = P.filt[j].off; h->filter_len[kk] = P.filt[j].len;
            uint8_t* d = h->alleles.data() + h->ref_off[kk];
            const unsigned char* r = reinterpret_cast<const unsigned char*>(base + P.ref[j].off);
            for (int32_t q = 0; q < P.ref[j].len; ++q) d[q] = code[r[q]];
            d = h->alleles.data() + h->alt_off[kk];
            const unsigned char* a = reinterpret_cast<const unsigned char*>(base + P.alt[j].off);
            for (int32_t q = 0; q < P.alt[j].len; ++q) d[q] = code[a[q]];
        }
    });
    st.lap("columns");
    *out = h.release();
    return 0;
}

int ugvc_vcf_part_info(const ugvc_vcf* h, int64_t* n_total, int64_t* part_lo) {
    if (!h) return fail("NULL argument");
    if (n_total) *n_total = h->n_total;
    if (part_lo) *part_lo = h->part_lo;
    return 0;
}

int ugvc_vcf_get_view(const ugvc_vcf* h, ugvc_vcf_view* v) {
    if (!h || !v) return fail("NULL argument");
    v->n = h->n;
    v->pool_bytes = (int64_t)h->alleles.size();
    v->contig = h->contig.data(); v->pos = h->pos.data();
    v->ref_len = h->ref_len.data(); v->alt_len = h->alt_len.data();
    v->ref_off = h->ref_off.data(); v->alt_off = h->alt_off.data();
    v->alleles = h->alleles.data();
    v->qual = h->qual.data(); v->sor = h->sor.data();
    v->dp = h->dp.data(); v->ad_ref = h->ad_ref.data(); v->ad_alt = h->ad_alt.data();
    v->gq = h->gq.data(); v->gt = h->gt.data(); v->tlod = h->tlod.data(); v->has_id = h->has_id.data();
    v->n_alt = h->n_alt.data(); v->rec_off = h->rec_off.data(); v->rec_len = h->rec_len.data();
    v->order = h->order.data();
    v->header = h->header_joined.data(); v->header_bytes = (int64_t)h->header_joined.size();
    v->text = h->text.data(); v->filter_off = h->filter_off.data(); v->filter_len = h->filter_len.data();
    return 0;
}

// (a std::thread that is still joinable when it is destroyed ends the process: the helper thread of the writer is joined on
// every way out of its scope, an exception included - ADVICE r4)
struct JoinedThread {
    std::thread t;
    ~JoinedThread() { if (t.joinable()) t.join(); }
};

static int write_filtered_impl(const ugvc_vcf* h, const char* out_path, const float* tree_score, const uint8_t* filter,
                               const uint8_t* flags, const uint8_t* cohort_extra, int64_t n, int n_threads, int write_index, FILE*& fh) {
    if (!h || !out_path || !tree_score || !filter || !flags) return fail("NULL argument");
    if (n != h->n) return fail("result columns do not match the record count of the input");
    if (h->n != h->n_total) return fail("this handle holds one part of the file (ugvc_vcf_read_part): the write-back needs the whole file");
    const int threads = pick_threads(n_threads);
    const char* base = h->text.data();
    const size_t plen = strlen(out_path);
    const bool gz = plen >= 3 && strcmp(out_path + plen - 3, ".gz") == 0;
    fh = fopen(out_path, "wb");
    if (!fh) return fail(std::string(out_path) + ": cannot open for writing");

    StageTimer st("write");
    // ---- header
    std::string stream;
    {
        std::vector<std::string_view> hdr, chrom;
        for (const Span& s : h->hdr_lines) {
            const std::string_view l(base + s.off, (size_t)s.len);
            (l.substr(0, 6) == "#CHROM" ? chrom : hdr).push_back(l);
        }
        auto have = [&](std::string_view key) {
            for (const Span& s : h->hdr_lines)
                if (std::string_view(base + s.off, (size_t)s.len).substr(0, key.size()) == key) return true;
            return false;
        };
        for (auto& l : hdr) { stream.append(l); stream.push_back('\n'); }
        for (const char* nh : kNewHeader) {
            const std::string_view full(nh);
            const std::string_view key = full.substr(0, full.find(','));
            if (!have(key)) { stream.append(full); stream.push_back('\n'); }
        }
        // the SEC filter line only when a row carries the tag (correct_systematic_errors): other outputs keep their bytes
        bool any_sec = false;
        for (int64_t k = 0; k < n && !any_sec; ++k) any_sec = (flags[k] & 4u) != 0;
        if (any_sec && !have("##FILTER=<ID=SEC")) {
            stream.append("##FILTER=<ID=SEC,Description=\"Systematic error: the cohort's allele counts explain the call\">");
            stream.push_back('\n');
        }
        for (auto& l : chrom) { stream.append(l); stream.push_back('\n'); }
    }
    std::vector<int64_t> row_of((size_t)n);
    for (int64_t k = 0; k < n; ++k) row_of[(size_t)h->order[(size_t)k]] = k;
    st.lap("header + row map");

    std::atomic<bool> io_ok{true};
    std::vector<uint32_t> blk_clen;                          // compressed size of every data block, file order
    std::vector<uint32_t> out_len((size_t)n);                // bytes of every output record line (with its newline)
    std::vector<int64_t> info_end(gz && write_index ? (size_t)n : 0, -1);   // INFO/END per record, for the index
    const size_t header_len = stream.size();
    static const unsigned char kEof[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0,
                                           0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    constexpr size_t kBlk = 65280;
    // emit every full 65280-byte block of data[0, size) (all of it when final); returns the bytes consumed
    // Round 6: a flush is TWO stages - the blocks of a batch are compressed (worker threads), then written in file order by a
    // writer thread of their own, while the NEXT batch is already being compressed into the other set of block buffers: format |
    // deflate | write run side by side (until round 5 a batch's write stood between its deflate and the next batch's).
    int deflate_threads = n_threads > 0 ? threads : (int)std::min<unsigned>(128u, std::max<unsigned>((unsigned)threads, std::thread::hardware_concurrency() / 2));
    if (const char* e = getenv("UGVC_VCF_DEFLATE_THREADS")) deflate_threads = std::max(1, atoi(e));       // (measurement knob)
    // compression level of the libdeflate back end (1..12; 6 = what htslib's bgzf writes by default, and the default here; the zlib back
    // end stays at 6 - its bytes are the pure-Python reference codec's).  Measured on 5 M records: profiles/r06_writer_sweep.txt
    int ld_level = 6;
    if (const char* e = getenv("UGVC_VCF_LEVEL")) ld_level = std::min(12, std::max(1, atoi(e)));
    struct CompPool {                                        // the threads' libdeflate compressors, kept across the flushes
        std::vector<void*> v;
        ~CompPool() { const LibDeflate& l = libdeflate(); for (void* c : v) if (c && l.ok) l.free_c(c); }
    } comp_pool;
    std::vector<void*>& ld_comp = comp_pool.v;
    std::vector<std::string> comp_buf[2];                    // compressed blocks of a flush, two sets (kept: their capacity is reused)
    int comp_sel = 0;
    std::atomic<int64_t> ns_deflate{0}, ns_write{0};         // (UGVC_VCF_TRACE: summed over the flushes)
    JoinedThread writer_guard;                               // (declared behind everything it touches: joined before those die)
    std::thread& writer = writer_guard.t;
    auto write_blocks = [&](const std::vector<std::string>* cv, size_t nb) {
        const auto t0 = std::chrono::steady_clock::now();
        try {
            for (size_t b = 0; b < nb; ++b) {
                const std::string& o = (*cv)[b];
                if (fwrite(o.data(), 1, o.size(), fh) != o.size()) io_ok = false;
                blk_clen.push_back((uint32_t)o.size());
            }
        } catch (...) { io_ok = false; }                     // (the writer thread: nothing thrown may leave it)
        ns_write += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    };
    auto flush_blocks = [&](const char* data, size_t size, bool final, bool timed) -> size_t {
        if (!gz) {
            if (size && fwrite(data, 1, size, fh) != size) io_ok = false;
            return size;
        }
        const size_t nb = final ? (size + kBlk - 1) / kBlk : size / kBlk;
        std::vector<std::string>& comp = comp_buf[comp_sel];
        if (comp.size() < nb) comp.resize(nb);
        const auto t_def0 = std::chrono::steady_clock::now();
        std::atomic<int> bad{0};
        // one deflate state per THREAD, reset from block to block: deflateInit2 allocates ~270 KB - above glibc's mmap
        // threshold, i.e. an mmap + 66 page faults + munmap per 64 KB block, serialised on the process' address-space lock
        // when 256 threads do it at once (0.3 s per 2 M records against 0.04 s of actual compression)
        std::atomic<int64_t> next_blk{0};
        // (round 6: the compression of a batch is what the writer waits for - 0.24-0.38 s of a 5 M-record write-back's 0.5 s with 64
        // threads, profiles/r06_c1_pipeline_5M.txt - so it takes up to half of the host's threads, 128 at most, and every thread
        // keeps its compressor from flush to flush instead of allocating one per flush)
        const int T = (int)std::max<int64_t>(1, std::min<int64_t>(deflate_threads, (int64_t)nb));
        const LibDeflate& ld = libdeflate();
        if (ld.ok && ld_comp.size() < (size_t)T) ld_comp.resize((size_t)T, nullptr);
        parallel_ranges(T, T, [&](int part_k, int64_t, int64_t) {
            static const unsigned char hd[16] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 'B', 'C', 0x02, 0};
            if (ld.ok) {
                void*& slot = ld_comp[(size_t)part_k];
                if (!slot) slot = ld.alloc_c(ld_level);
                void* c = slot;
                if (!c) { bad = 1; return; }
                const size_t bound = ld.bound(c, kBlk);
                for (;;) {
                    const int64_t b = next_blk.fetch_add(1);
                    if (b >= (int64_t)nb) break;
                    const size_t lo = (size_t)b * kBlk, len = std::min(kBlk, size - lo);
                    std::string& o = comp[(size_t)b];
                    o.resize(18 + bound + 8);
                    const size_t clen = ld.compress(c, data + lo, len, &o[18], o.size() - 26);
                    if (clen == 0 || clen + 25 > 65535) { bad = 1; break; }
                    memcpy(&o[0], hd, 16);
                    const uint32_t bsize = (uint32_t)(clen + 25);
                    o[16] = (char)(bsize & 0xff); o[17] = (char)(bsize >> 8);
                    const uint32_t crc = ld.crc(0, data + lo, len);
                    unsigned char* t = reinterpret_cast<unsigned char*>(&o[18 + clen]);
                    for (int i = 0; i < 4; ++i) { t[i] = (unsigned char)(crc >> (8 * i)); t[4 + i] = (unsigned char)((uint32_t)len >> (8 * i)); }
                    o.resize(18 + clen + 8);
                }
                return;
            }
            z_stream zs;
            memset(&zs, 0, sizeof zs);
            if (deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { bad = 1; return; }
            const size_t bound = deflateBound(&zs, (uLong)kBlk);
            for (;;) {
                const int64_t b = next_blk.fetch_add(1);
                if (b >= (int64_t)nb) break;
                const size_t lo = (size_t)b * kBlk, len = std::min(kBlk, size - lo);
                if (deflateReset(&zs) != Z_OK) { bad = 1; break; }
                std::string& o = comp[(size_t)b];
                o.resize(18 + bound + 8);
                zs.next_in = reinterpret_cast<unsigned char*>(const_cast<char*>(data + lo));
                zs.avail_in = (uInt)len;
                zs.next_out = reinterpret_cast<unsigned char*>(&o[18]);
                zs.avail_out = (uInt)(o.size() - 26);
                const int rc = deflate(&zs, Z_FINISH);
                const size_t clen = zs.total_out;
                if (rc != Z_STREAM_END || clen + 25 > 65535) { bad = 1; break; }
                memcpy(&o[0], hd, 16);
                const uint32_t bsize = (uint32_t)(clen + 25);
                o[16] = (char)(bsize & 0xff); o[17] = (char)(bsize >> 8);
                const uint32_t crc = (uint32_t)crc32(0L, reinterpret_cast<const unsigned char*>(data + lo), (uInt)len);
                unsigned char* t = reinterpret_cast<unsigned char*>(&o[18 + clen]);
                for (int i = 0; i < 4; ++i) { t[i] = (unsigned char)(crc >> (8 * i)); t[4 + i] = (unsigned char)((uint32_t)len >> (8 * i)); }
                o.resize(18 + clen + 8);
            }
            deflateEnd(&zs);
        });
        ns_deflate += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_def0).count();
        if (bad) { io_ok = false; return 0; }
        if (timed) st.lap("  deflate");
        // hand the blocks to the writer: the previous flush's write (the other buffer set) must be through - blocks reach the file
        // in order, one write at a time - and runs beside the next flush's deflate
        if (writer.joinable()) writer.join();
        writer = std::thread(write_blocks, &comp, nb);
        comp_sel ^= 1;
        if (final) writer.join();
        if (timed) st.lap("  file write");
        return std::min(size, nb * kBlk);
    };

    // ---- records, in batches of file-order ranges
    // (batches of 512 k records, ~90 MB of text: every batch starts its threads twice - 19 batches of 256 k records with 256
    // threads each were ~10 000 thread starts, a third of the writer's time in round 2; with 64 threads and the compression of
    // a batch running beside the next batch's formatting, ten batches of a 5 M-record file overlap better than three)
    int64_t batch = 1 << 19;
    if (const char* e = getenv("UGVC_VCF_WRITE_BATCH")) batch = std::max<int64_t>(1, atoll(e));      // (tests: batch seams on small files)
    // A batch's blocks are compressed and written by ONE helper thread (with its own worker threads) while the next batch is
    // formatted: what a batch leaves over for the next one - less than a block - is the tail of its text, known before a byte of
    // it is compressed.  (Round 4: format + gather were 0.17 s of the 5 M-record write-back's 0.45 s, in front of 0.26 s of deflate.)
    // (the per-thread text buffers and the two gather buffers live across the batches: fresh ones were ~90 MB of first-touch
    // page faults per batch, taken under the address-space lock that the compressing threads' allocations want too)
    std::vector<std::string> part;
    std::unique_ptr<char[]> gbuf[2];
    size_t gcap[2] = {0, 0};
    int gsel = 0;
    // (declared AFTER everything the flusher reads: locals unwind in reverse order, so on an exception - a bad_alloc for the next
    // batch's buffer - the guard joins the helper thread BEFORE the gather buffers it is deflating from are freed)
    JoinedThread flusher_guard;
    std::thread& flusher = flusher_guard.t;
    for (int64_t b0 = 0; b0 < n && io_ok; b0 += batch) {
        const int64_t b1 = std::min(n, b0 + batch);
        const int parts = (int)std::max<int64_t>(1, std::min<int64_t>(threads, (b1 - b0) / 1024));
        if (part.size() < (size_t)parts) part.resize((size_t)parts);
        parallel_ranges(b1 - b0, parts, [&](int p, int64_t lo, int64_t hi) {
            std::string& o = part[(size_t)p];
            o.clear();
            o.reserve((size_t)(hi - lo) * 160);
            char num[64];
            for (int64_t j = b0 + lo; j < b0 + hi; ++j) {
                const Span ln = h->rec_lines[(size_t)j];
                const size_t k = (size_t)row_of[(size_t)j];
                const size_t o_before = o.size();
                const char* s = base + ln.off;
                const char* e = s + ln.len;
                const char* tab[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
                int nt = 0;
                for (const char* q = s; nt < 8;) {
                    const char* t = static_cast<const char*>(memchr(q, '\t', (size_t)(e - q)));
                    if (!t) break;
                    tab[nt++] = t;
                    q = t + 1;
                }
                // fields 6 and 7 lie between tab[5]..tab[6] and tab[6]..(tab[7] | e); read validated >= 8 fields
                if (nt < 7) continue;
                const char* f6 = tab[5] + 1;
                const char* f7 = tab[6] + 1;
                const char* f7e = nt >= 8 ? tab[7] : e;
                if (!info_end.empty()) info_end[(size_t)j] = info_end_of(f7, f7e);
                o.append(s, (size_t)(f6 - s));
                const unsigned fl = flags[k];
                bool any = false;
                if (fl & 1u) { o.append("HPOL_RUN"); any = true; }
                if ((fl & 2u) || (cohort_extra && cohort_extra[k])) { if (any) o.push_back(';'); o.append("COHORT_FP"); any = true; }
                if (fl & 4u) { if (any) o.push_back(';'); o.append("SEC"); any = true; }
                if (filter[k] == 1) { if (any) o.push_back(';'); o.append("LOW_SCORE"); any = true; }
                if (!any) o.append("PASS");
                o.push_back('\t');
                bool wrote = false;
                if (!((f7e - f7 == 1 && *f7 == '.') || f7e == f7)) {
                    const char* q = f7;
                    for (;;) {
                        const char* t = static_cast<const char*>(memchr(q, ';', (size_t)(f7e - q)));
                        const char* fe = t ? t : f7e;
                        const size_t len = (size_t)(fe - q);
                        const bool drop = (len >= 11 && memcmp(q, "TREE_SCORE=", 11) == 0) || (len == 8 && memcmp(q, "HPOL_RUN", 8) == 0);
                        if (!drop) {
                            if (wrote) o.push_back(';');
                            o.append(q, len);
                            wrote = true;
                        }
                        if (!t) break;
                        q = t + 1;
                    }
                }
                if (wrote) o.push_back(';');
                o.append("TREE_SCORE=");
                const int nn = format_f32(tree_score[k], num, (int)sizeof num);
                o.append(num, (size_t)(nn > 0 ? nn : 0));
                if (fl & 1u) o.append(";HPOL_RUN");
                o.append(f7e, (size_t)(e - f7e));
                o.push_back('\n');
                out_len[(size_t)j] = (uint32_t)(o.size() - o_before);
            }
        });
        // what is left of the previous batch (less than a block) and the parts go to their places in ONE uninitialised
        // buffer, in parallel (a serial append of ~45 MB per batch was a fifth of the writer's time at 5 M records;
        // std::string::resize would zero-fill the buffer first)
        st.lap("format");
        std::vector<size_t> at((size_t)parts + 1, stream.size());
        for (int p = 0; p < parts; ++p) at[(size_t)p + 1] = at[(size_t)p] + part[(size_t)p].size();
        const size_t total = at[(size_t)parts];
        // (the buffer of the batch before the previous one: its flush was joined before the previous batch's was started)
        if (gcap[gsel] < total) { gbuf[gsel].reset(new char[total + total / 8 + 1]); gcap[gsel] = total + total / 8 + 1; }
        char* dst = gbuf[gsel].get();
        memcpy(dst, stream.data(), stream.size());
        parallel_ranges(parts, std::min(parts, threads), [&](int, int64_t lo, int64_t hi) {
            for (int64_t p = lo; p < hi; ++p) memcpy(dst + at[(size_t)p], part[(size_t)p].data(), part[(size_t)p].size());
        });
        st.lap("gather parts");
        if (flusher.joinable()) flusher.join();                 // (blocks reach the file in order: one flush at a time)
        st.lap("wait for the previous batch's deflate + write");
        const size_t used = gz ? total / kBlk * kBlk : total;
        stream.assign(dst + used, total - used);
        // (nothing thrown on the helper thread may leave it - an allocation of a block buffer, a thread that cannot start: the write
        // is marked failed and the call returns an error instead of the process ending in std::terminate)
        flusher = std::thread([&flush_blocks, &io_ok, dst, total] {
            try { (void)flush_blocks(dst, total, false, false); } catch (...) { io_ok = false; }
        });
        gsel ^= 1;
    }
    if (flusher.joinable()) flusher.join();
    st.lap("last batch's deflate");
    if (io_ok) (void)flush_blocks(stream.data(), stream.size(), true, true);
    if (writer.joinable()) writer.join();
    if (st.on) fprintf(stderr, "[vcf] write   (all flushes: deflate %.4f s, file write %.4f s - side by side)\n", ns_deflate.load() * 1e-9, ns_write.load() * 1e-9);
    if (io_ok && gz && fwrite(kEof, 1, 28, fh) != 28) io_ok = false;
    if (fclose(fh) != 0) io_ok = false;
    fh = nullptr;
    if (!io_ok) return fail(std::string(out_path) + ": write failed");
    st.lap("tail + close");
    if (gz && write_index) {
        const int rc = write_tbi(h, out_path, row_of, out_len, info_end, header_len, blk_clen, threads);
        st.lap("tabix index");
        return rc;
    }
    return 0;
}

int ugvc_vcf_write_filtered(const ugvc_vcf* h, const char* out_path, const float* tree_score, const uint8_t* filter,
                            const uint8_t* flags, const uint8_t* cohort_extra, int64_t n, int n_threads, int write_index) {
    // nothing thrown inside (an allocation of a batch buffer, a thread that cannot start) crosses the C boundary
    FILE* fh = nullptr;
    int rc;
    try {
        rc = write_filtered_impl(h, out_path, tree_score, filter, flags, cohort_extra, n, n_threads, write_index, fh);
    } catch (const std::exception& e) {
        rc = fail(std::string("write-back failed: ") + e.what());
    } catch (...) {
        rc = fail("write-back failed: unknown exception");
    }
    if (fh) (void)fclose(fh);                                    // (left open by an error or an exception)
    return rc;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------
// Side tables: FASTA -> base codes, BED / interval_list -> (contig, start, end) rows
// ------------------------------------------------------------------------------------------------------------
struct ugvc_fasta {
    std::vector<uint8_t, NoInitAlloc<uint8_t>> codes;         // (every byte is written by the encode pass: no zero-fill of 3 GB first)
    std::vector<int64_t> off;
    std::string names;
};

struct ugvc_intervals {
    std::vector<int64_t> contig, start, end;
    // the finished track (ugvc_intervals_track): sorted, empty rows dropped, merged if asked; CSR pointers per contig
    std::vector<int32_t> t_start, t_end, t_ptr;
};

extern "C" {

int ugvc_fasta_read(const char* path, int n_threads, ugvc_fasta** out) {
    if (!path || !out) return fail("NULL argument");
    *out = nullptr;
    const int threads = pick_threads(n_threads);
    StageTimer st("fasta");
    TextBuf text;
    if (load_text(path, threads, text)) return -1;
    st.lap("load");
    const char* base = text.data();
    const int64_t tn = (int64_t)text.size();
    if (tn == 0 || base[0] != '>') return fail(std::string(path) + ": not a FASTA file");
    // headers: every line that starts with '>'
    std::vector<int64_t> hdr;                       // offset of '>'
    {
        const int parts = (int)std::max<int64_t>(1, std::min<int64_t>(threads, tn / (1 << 20)));
        std::vector<std::vector<int64_t>> found((size_t)parts);
        parallel_ranges(tn, parts, [&](int p, int64_t lo, int64_t hi) {
            for (int64_t q = lo; q < hi;) {
                const void* g = memchr(base + q, '>', (size_t)(hi - q));
                if (!g) break;
                const int64_t at = (const char*)g - base;
                if (at == 0 || base[at - 1] == '\n') found[(size_t)p].push_back(at);
                q = at + 1;
            }
        });
        for (auto& f : found) hdr.insert(hdr.end(), f.begin(), f.end());
    }
    const size_t nrec = hdr.size();
    std::unique_ptr<ugvc_fasta> h(new ugvc_fasta());
    std::vector<int64_t> seq_lo(nrec), seq_hi(nrec);
    for (size_t k = 0; k < nrec; ++k) {
        const void* nl = memchr(base + hdr[k], '\n', (size_t)(tn - hdr[k]));
        const int64_t e = nl ? (const char*)nl - base : tn;
        // name: first whitespace-delimited token after '>'
        int64_t a = hdr[k] + 1, b = a;
        while (a < e && is_space(base[a])) ++a;
        b = a;
        while (b < e && !is_space(base[b])) ++b;
        if (k) h->names.push_back('\n');
        h->names.append(base + a, (size_t)(b - a));
        seq_lo[k] = std::min(e + 1, tn);
        seq_hi[k] = k + 1 < nrec ? hdr[k + 1] : tn;
    }
    // work items: <= 2 MB pieces of every record's sequence range; count, prefix, encode
    struct Piece { int64_t lo, hi, out; size_t rec; };
    std::vector<Piece> pieces;
    for (size_t k = 0; k < nrec; ++k)
        for (int64_t q = seq_lo[k]; q < seq_hi[k] || q == seq_lo[k]; q += (int64_t)2 << 20) {
            pieces.push_back(Piece{q, std::min(seq_hi[k], q + ((int64_t)2 << 20)), 0, k});
            if (q + ((int64_t)2 << 20) >= seq_hi[k]) break;
        }
    std::vector<int64_t> cnt(pieces.size());
    parallel_items((int64_t)pieces.size(), threads, [&](int64_t i) {
        const Piece& p = pieces[(size_t)i];
        int64_t c = 0;
        for (int64_t q = p.lo; q < p.hi; ++q) c += base[q] != '\n' && base[q] != '\r';
        cnt[(size_t)i] = c;
    });
    h->off.assign(nrec + 1, 0);
    int64_t run = 0;
    for (size_t i = 0; i < pieces.size(); ++i) {
        pieces[i].out = run;
        run += cnt[i];
        h->off[pieces[i].rec + 1] = run;
    }
    for (size_t k = 1; k <= nrec; ++k) h->off[k] = std::max(h->off[k], h->off[k - 1]);   // records without sequence
    st.lap("headers + count");
    h->codes.resize((size_t)run);
    uint8_t code[256];
    memset(code, 0, sizeof code);
    code['A'] = code['a'] = 1; code['C'] = code['c'] = 2; code['G'] = code['g'] = 3; code['T'] = code['t'] = 4;
    parallel_items((int64_t)pieces.size(), threads, [&](int64_t i) {
        const Piece& p = pieces[(size_t)i];
        uint8_t* d = h->codes.data() + p.out;
        for (int64_t q = p.lo; q < p.hi; ++q) {
            const unsigned char ch = (unsigned char)base[q];
            if (ch != '\n' && ch != '\r') *d++ = code[ch];
        }
    });
    st.lap("encode");
    *out = h.release();
    return 0;
}

int ugvc_fasta_get_view(const ugvc_fasta* h, ugvc_fasta_view* v) {
    if (!h || !v) return fail("NULL argument");
    v->total = (int64_t)h->codes.size();
    v->n_contigs = (int32_t)(h->off.size() - 1);
    v->codes = h->codes.data();
    v->contig_off = h->off.data();
    v->names = h->names.data();
    v->names_bytes = (int64_t)h->names.size();
    return 0;
}

void ugvc_fasta_free(ugvc_fasta* h) { delete h; }

int ugvc_intervals_read(const char* path, const char* const* contig_names, int n_contigs, int n_threads, ugvc_intervals** out) {
    if (!path || !out || (n_contigs > 0 && !contig_names)) return fail("NULL argument");
    *out = nullptr;
    const int threads = pick_threads(n_threads);
    TextBuf text;
    if (load_text(path, threads, text)) return -1;
    const char* base = text.data();
    const int64_t tn = (int64_t)text.size();
    std::vector<std::string> names((size_t)std::max(n_contigs, 0));
    std::unordered_map<std::string_view, int> idx;
    for (int c = 0; c < n_contigs; ++c) names[(size_t)c] = contig_names[c] ? contig_names[c] : "";
    for (int c = 0; c < n_contigs; ++c) idx[std::string_view(names[(size_t)c])] = c;
    const size_t plen = strlen(path);
    const bool by_ext = plen >= 14 && strcmp(path + plen - 14, ".interval_list") == 0;
    // one-based from the first '@' line on (Picard header), or for the whole file by extension
    int64_t one_from = by_ext ? 0 : INT64_MAX;
    for (int64_t s = 0; s < tn && !by_ext;) {
        const void* nl = memchr(base + s, '\n', (size_t)(tn - s));
        const int64_t e = nl ? (const char*)nl - base : tn;
        if (base[s] == '@') { one_from = s; break; }
        s = e + 1;
    }
    const int parts = (int)std::max<int64_t>(1, std::min<int64_t>(threads, tn / (1 << 18)));
    std::vector<ugvc_intervals> part((size_t)parts);
    std::vector<std::string> errs((size_t)parts);
    parallel_ranges(tn, parts, [&](int p, int64_t lo, int64_t hi) {
        int64_t s = lo;
        if (lo > 0) {
            const void* nl = memchr(base + lo - 1, '\n', (size_t)(tn - lo + 1));
            if (!nl) return;
            s = (const char*)nl - base + 1;
        }
        ugvc_intervals& o = part[(size_t)p];
        while (s < hi && s < tn) {
            const void* nl = memchr(base + s, '\n', (size_t)(tn - s));
            const int64_t e = nl ? (const char*)nl - base : tn;
            const int64_t ls = s;
            s = e + 1;
            bool blank = true;
            for (int64_t q = ls; q < e && blank; ++q) blank = is_space(base[q]);
            if (blank || base[ls] == '#' || base[ls] == '@') continue;
            if ((e - ls >= 5 && memcmp(base + ls, "track", 5) == 0) || (e - ls >= 7 && memcmp(base + ls, "browser", 7) == 0)) continue;
            // fields: tab-separated; with fewer than three tab fields, any whitespace separates
            Span f[3];
            int nf = 0;
            {
                const char* q = base + ls;
                const char* le = base + e;
                int tabs = 0;
                for (const char* t = q; t < le; ++t) tabs += *t == '\t';
                if (tabs >= 2) {
                    for (; nf < 3; ++nf) {
                        const char* t = static_cast<const char*>(memchr(q, '\t', (size_t)(le - q)));
                        const char* fe = t ? t : le;
                        f[nf] = Span{(int64_t)(q - base), (int32_t)(fe - q)};
                        if (!t) { ++nf; break; }
                        q = t + 1;
                    }
                } else {
                    while (nf < 3) {
                        while (q < le && is_space(*q)) ++q;
                        if (q >= le) break;
                        const char* fe = q;
                        while (fe < le && !is_space(*fe)) ++fe;
                        f[nf++] = Span{(int64_t)(q - base), (int32_t)(fe - q)};
                        q = fe;
                    }
                }
            }
            if (nf < 3) { if (errs[(size_t)p].empty()) errs[(size_t)p] = std::string(path) + ": interval line with fewer than 3 columns"; return; }
            auto it = idx.find(std::string_view(base + f[0].off, (size_t)f[0].len));
            if (it == idx.end()) continue;               // contig not in the reference: cannot annotate any variant
            int64_t a, b;
            if (!parse_int(base + f[1].off, f[1].len, a) || !parse_int(base + f[2].off, f[2].len, b)) {
                if (errs[(size_t)p].empty()) errs[(size_t)p] = std::string(path) + ": interval coordinates are not integers";
                return;
            }
            o.contig.push_back(it->second);
            o.start.push_back(a - (ls >= one_from ? 1 : 0));
            o.end.push_back(b);
        }
    });
    for (auto& e : errs) if (!e.empty()) return fail(e);
    std::unique_ptr<ugvc_intervals> h(new ugvc_intervals());
    for (auto& o : part) {
        h->contig.insert(h->contig.end(), o.contig.begin(), o.contig.end());
        h->start.insert(h->start.end(), o.start.begin(), o.start.end());
        h->end.insert(h->end.end(), o.end.begin(), o.end.end());
    }
    *out = h.release();
    return 0;
}

int ugvc_intervals_get_view(const ugvc_intervals* h, ugvc_intervals_view* v) {
    if (!h || !v) return fail("NULL argument");
    v->n = (int64_t)h->contig.size();
    v->contig = h->contig.data();
    v->start = h->start.data();
    v->end = h->end.data();
    return 0;
}

// The rows as the engine wants them (variantcalling_amd/io/bed.py: track_from_arrays is the statement this follows, line by line;
// tests/test_vcf_native.py compares the two on sorted, shuffled, nested, book-ended and empty inputs): ordered by (contig, start,
// end) - checked first, a BED file is sorted as a rule -, rows with end <= start dropped, with `merge` a new interval opens where
// a start lies beyond the running maximum of the ends seen so far in its contig, 32-bit starts / ends, row range per contig.
// (Round 5 did this in numpy on the reader's int64 columns: 0.2 of the 0.4 s the CLI's first stage waited for a 3 M-row track.)
int ugvc_intervals_track(ugvc_intervals* h, int n_contigs, int merge, ugvc_track_view* v) {
    if (!h || !v || n_contigs < 0) return fail("NULL argument");
    try {
        const size_t n = h->contig.size();
        const int64_t* c = h->contig.data();
        const int64_t* s = h->start.data();
        const int64_t* e = h->end.data();
        bool in_order = true;
        for (size_t i = 1; i < n && in_order; ++i)
            in_order = c[i] > c[i - 1] || (c[i] == c[i - 1] && (s[i] > s[i - 1] || (s[i] == s[i - 1] && e[i] >= e[i - 1])));
        std::vector<uint32_t> order;
        if (!in_order) {
            if (n > 0xFFFFFFFFull) return fail("interval file too large");
            order.resize(n);
            for (size_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
            std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
                if (c[x] != c[y]) return c[x] < c[y];
                if (s[x] != s[y]) return s[x] < s[y];
                return e[x] < e[y];
            });
        }
        h->t_start.resize(n); h->t_end.resize(n);
        h->t_ptr.assign((size_t)n_contigs + 1, 0);
        int32_t* const os = h->t_start.data();
        int32_t* const oe = h->t_end.data();
        std::vector<int64_t> per_contig((size_t)n_contigs + 1, 0);           // rows kept per contig
        int64_t cur_c = -1, cur_max = 0;
        size_t m = 0;                                                        // rows written
        for (size_t k = 0; k < n; ++k) {
            const size_t i = in_order ? k : (size_t)order[k];
            const int64_t si = s[i], ei = e[i], ci = c[i];
            if (!(ei > si)) continue;
            // a new interval: first row, another contig, no merging, or a start beyond every end seen so far in this contig
            if (m == 0 || ci != cur_c || !merge || si > cur_max) {
                os[m] = (int32_t)si;
                oe[m] = (int32_t)ei;
                ++m;
                if (ci >= 0 && ci < n_contigs) ++per_contig[(size_t)ci];
                cur_max = ei;                                                // (same contig and merging: s > every end so far, so e is the new maximum)
                cur_c = ci;
            } else if (ei > cur_max) {
                cur_max = ei;
                oe[m - 1] = (int32_t)ei;
            }
        }
        h->t_start.resize(m); h->t_end.resize(m);
        for (int q = 0; q < n_contigs; ++q) h->t_ptr[(size_t)q + 1] = h->t_ptr[(size_t)q] + (int32_t)per_contig[(size_t)q];
        v->n = (int64_t)h->t_start.size();
        v->start = h->t_start.data();
        v->end = h->t_end.data();
        v->ptr = h->t_ptr.data();
        return 0;
    } catch (const std::exception& ex) {
        return fail(std::string("interval track: ") + ex.what());
    }
}

void ugvc_intervals_free(ugvc_intervals* h) { delete h; }

}  // extern "C"
