// The host-buffer boundary made fast: ugvc_filter_variants as a chunk pipeline (round 3).
//
// What the call replaces is the reference's per-record loop around the model (get_vcf_df ... predict ... write:
// /root/reference/ugvc/reports/report_wo_gt.ipynb:1207-1210); what round 2 shipped was upload-everything, one pass,
// download-everything from PAGEABLE caller memory with a serial 5 M-iteration validation loop in front: 10.2 ms for a
// callset the resident pass scores in 0.5 ms.  Here the callset is cut into K row chunks and three things overlap:
//   host     a small pool of threads validates a chunk's rows (sortedness, contig range, allele bounds - what the
//            kernels rely on) WHILE packing its twelve columns back to back into a pinned slot (two slots, ping-pong);
//   H2D      ONE DMA per chunk moves the slot to a device staging block (the host link moves one 200 MB copy at 57.5 GB/s
//            but 96 pieces of 2 MB at 45: tools/calib/pcie_probe.hip, profiles/r03_pcie_probe.txt); twelve device copies
//            at HBM rate put the columns in their place in the resident columns;
//   compute  the scoring pass over that row range runs on the context stream behind an event, and its three result
//            columns come back by DMA on a third stream into pinned memory, from where the pool copies them into the
//            caller's arrays while later chunks are still in flight.
// Rows are independent given the resident side tables, so a chunk is scored exactly as it would be
// inside the whole callset; afterwards the context is in the same state as after upload + ugvc_filter_resident
// (columns and results resident, `scored` set).
#include <sched.h>
#include <unistd.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <fstream>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include "host_pool.hpp"
#include "ugvc_v2.hpp"

namespace ugvc {

int launch_score(ugvc_ctx* ctx, const FilterArgs& a);
// host_rows.cpp (g++, vectorised)
int validate_rows(const ugvc_variants* v, int64_t lo, int64_t hi, int n_contigs, int64_t* n_indel, int64_t* row);
const char* row_error_text(int what);
bool offsets_canonical(const ugvc_variants* v, int64_t lo, int64_t hi, int64_t first);
void copy_stream(void* dst, const void* src, size_t n);
void copy_stream_fence();

// CPUs of the NUMA node the GPU hangs on (sysfs: the PCI device's numa_node, the node's cpulist); false when the host has one
// node, hides the topology, or UGVC_NO_NUMA is set.  On the two-socket hosts of this pool the boundary call takes 5.3-5.6 ms
// with pool and staging on the GPU's socket and 7-12 ms when the scheduler happens to put them on the other one.
static bool gpu_node_cpus(int device, cpu_set_t& out) {
    if (getenv("UGVC_NO_NUMA")) return false;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, sizeof(bus), device) != hipSuccess) return false;
    for (char* c = bus; *c; ++c) *c = (char)tolower(*c);
    int node = -1;
    {
        std::ifstream f(std::string("/sys/bus/pci/devices/") + bus + "/numa_node");
        if (!(f >> node) || node < 0) return false;
    }
    std::ifstream f("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist");
    std::string list;
    if (!std::getline(f, list) || list.empty()) return false;
    CPU_ZERO(&out);
    int n_set = 0;
    size_t at = 0;
    while (at < list.size()) {
        size_t end = list.find(',', at);
        if (end == std::string::npos) end = list.size();
        const std::string part = list.substr(at, end - at);
        const size_t dash = part.find('-');
        const int lo = atoi(part.c_str()), hi = dash == std::string::npos ? lo : atoi(part.c_str() + dash + 1);
        for (int c = lo; c <= hi && c < CPU_SETSIZE; ++c) { CPU_SET(c, &out); ++n_set; }
        at = end + 1;
    }
    return n_set > 0;
}

struct PipeState {
    static constexpr int kSlots = 3;            // staging slots in flight: being packed | in the DMA | being read by a pass
    HostPool* pool = nullptr;
    bool numa_known = false, numa = false;      // the GPU's NUMA node, looked up once
    cpu_set_t node_cpus;
    hipStream_t h2d = nullptr, d2h = nullptr;
    void* stage[kSlots] = {nullptr, nullptr, nullptr};   // pinned: one chunk's columns, back to back
    size_t stage_cap[kSlots] = {0, 0, 0};
    DeviceBuf d_stage[kSlots];                  // where a slot lands on the device; the chunk's pass reads its columns HERE
    void* res = nullptr;                        // pinned: every chunk's packed result columns
    size_t res_cap = 0;
    DeviceBuf d_res;                            // the passes write their results HERE, packed per chunk
    DeviceBuf d_sums[kSlots];                   // allele-offset reconstruction: one sum per 4096 rows of a chunk
    void* alle = nullptr;                       // pinned: the allele pool
    size_t alle_cap = 0;
    std::vector<hipEvent_t> ev;                 // four per chunk
    bool ev_timed = false;                      // (UGVC_PIPE_TRACE: the events carry timestamps)
    hipEvent_t ev_t0 = nullptr;
};

static PipeState* pipe_state(ugvc_ctx* ctx) {
    if (!ctx->pipe) ctx->pipe = new PipeState();
    return static_cast<PipeState*>(ctx->pipe);
}

void pipe_destroy(ugvc_ctx* ctx) {
    if (!ctx->pipe) return;
    PipeState* p = static_cast<PipeState*>(ctx->pipe);
    delete p->pool;
    for (void* q : {p->stage[0], p->stage[1], p->stage[2], p->res, p->alle})
        if (q) (void)hipHostFree(q);
    for (DeviceBuf* b : {&p->d_stage[0], &p->d_stage[1], &p->d_stage[2], &p->d_res, &p->d_sums[0], &p->d_sums[1], &p->d_sums[2]})
        if (b->p) dev_free(b->p);
    for (hipEvent_t e : p->ev) (void)hipEventDestroy(e);
    if (p->ev_t0) (void)hipEventDestroy(p->ev_t0);
    if (p->h2d) (void)hipStreamDestroy(p->h2d);
    if (p->d2h) (void)hipStreamDestroy(p->d2h);
    delete p;
    ctx->pipe = nullptr;
}

static int pinned(void*& p, size_t& cap, size_t bytes) {
    if (bytes <= cap && p) return 0;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    UGVC_HIP(hipHostMalloc(&p, std::max<size_t>(bytes, 4096), hipHostMallocDefault));
    cap = std::max<size_t>(bytes, 4096);
    return 0;
}

// the columns of ugvc_variants, in the order they sit in a staging slot
struct Col {
    const void* src;
    DeviceBuf* dst;
    size_t w;
};

// ---- one launch moves a chunk's fifteen segments (twelve columns out of the staging block, three result columns out of the
// packed result block) into the resident columns - off the critical path: the pass has already read the staging block and
// the results are already on their way to the host.  Every segment starts 16-byte aligned on both sides (chunks start at
// multiples of 64 rows, slot offsets are multiples of 64 bytes).
struct Seg {
    const uint8_t* src;
    uint8_t* dst;
    uint64_t bytes;
};
struct SegTable {
    Seg s[16];
};

__global__ void __launch_bounds__(256) pipe_place_kernel(SegTable t) {
    const Seg s = t.s[blockIdx.y];
    const uint64_t n16 = s.bytes >> 4;
    const uint4* __restrict__ src = reinterpret_cast<const uint4*>(s.src);
    uint4* __restrict__ dst = reinterpret_cast<uint4*>(s.dst);
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) dst[i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x < (unsigned)(s.bytes & 15)) s.dst[(n16 << 4) + threadIdx.x] = s.src[(n16 << 4) + threadIdx.x];
}

// ---- allele offsets that are not shipped.  When a chunk's pool is laid out canonically (host_rows.cpp: every row's REF
// then ALT, row after row - what a VCF reader leaves behind) ref_off / alt_off are an exclusive scan of ref_len + alt_len
// from the chunk's first offset: 8 of a row's 39 bytes stay off the host link.  Two small launches per chunk, in front of
// its pass: sums of 4096-row blocks, then every block adds the sums in front of it to its own scan.  u32 arithmetic as in
// the columns themselves (a pool is at most 4 GiB).
constexpr int kOffRows = 16;                                     // rows per thread; a block of 256 threads covers 4096 rows

__global__ void __launch_bounds__(256) pipe_offsets_sum_kernel(const uint16_t* __restrict__ rl, const uint16_t* __restrict__ al, int64_t m,
                                                               uint32_t* __restrict__ sums) {
    __shared__ uint32_t part[4];
    const int64_t r0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * kOffRows;
    uint32_t s = 0;
    for (int q = 0; q < kOffRows; ++q)
        if (r0 + q < m) s += (uint32_t)rl[r0 + q] + al[r0 + q];
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

__global__ void __launch_bounds__(256) pipe_offsets_fill_kernel(const uint16_t* __restrict__ rl, const uint16_t* __restrict__ al, int64_t m,
                                                                uint32_t base, const uint32_t* __restrict__ sums, uint32_t* __restrict__ ro,
                                                                uint32_t* __restrict__ ao) {
    __shared__ uint32_t wave_tot[4];
    __shared__ uint32_t before;
    // the blocks in front of this one (a chunk has at most a few hundred)
    uint32_t pre = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) pre += sums[b];
    for (int d = 32; d >= 1; d >>= 1) pre += __shfl_xor(pre, d);
    if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = pre;
    __syncthreads();
    if (threadIdx.x == 0) before = base + wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    __syncthreads();
    const int64_t r0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * kOffRows;
    uint32_t len[kOffRows], mine = 0;
    for (int q = 0; q < kOffRows; ++q) {
        len[q] = r0 + q < m ? (uint32_t)rl[r0 + q] + al[r0 + q] : 0u;
        mine += len[q];
    }
    // exclusive scan of the threads' totals: inside the wave by shuffles, across the four waves through LDS
    uint32_t incl = mine;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    __syncthreads();                                             // (wave_tot is reused)
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t at = before + incl - mine;
    for (int w = 0; w < wave; ++w) at += wave_tot[w];
    for (int q = 0; q < kOffRows; ++q)
        if (r0 + q < m) {
            ro[r0 + q] = at;
            ao[r0 + q] = at + rl[r0 + q];
            at += len[q];
        }
}

int filter_variants_pipelined(ugvc_ctx* ctx, const ugvc_variants* v, const ugvc_results* out, int n_chunks, bool reserve_only) {
    // UGVC_PIPE_TRACE=1: host and device timeline of the call on stderr (ms since entry; tools/pipe_trace.py reads it)
    static const bool trace = getenv("UGVC_PIPE_TRACE") != nullptr;
    const auto t_entry = std::chrono::steady_clock::now();
    auto now_ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_entry).count(); };
    struct HostMark { int chunk; const char* what; double ms; };
    std::vector<HostMark> marks;
    auto mark = [&](int c, const char* what) { if (trace) marks.push_back({c, what, now_ms()}); };

    constexpr int NS = PipeState::kSlots;
    const int64_t n = v->n;
    PipeState* ps = pipe_state(ctx);
    // pool and pinned staging live on the GPU's NUMA node: the calling thread is moved there for the duration of the call
    // (it hands the chunks over) and put back on its own CPUs afterwards
    if (!ps->numa_known) {
        ps->numa = gpu_node_cpus(ctx->device, ps->node_cpus);
        ps->numa_known = true;
    }
    // (a launcher's or user's CPU binding wins: when the calling thread's affinity mask is already restricted - taskset, a
    // per-rank core binding - neither it nor the pool threads are moved; UGVC_NO_NUMA=1 switches the placement off)
    cpu_set_t mine;
    bool numa = ps->numa && sched_getaffinity(0, sizeof(mine), &mine) == 0;
    if (numa) {
        const long online = sysconf(_SC_NPROCESSORS_ONLN);
        cpu_set_t inter;
        CPU_AND(&inter, &mine, &ps->node_cpus);
        if ((online > 0 && CPU_COUNT(&mine) < online) || CPU_COUNT(&inter) == 0) numa = false;
    }
    struct Restore {
        bool on; cpu_set_t set;
        ~Restore() { if (on) (void)sched_setaffinity(0, sizeof(set), &set); }
    } restore{numa, mine};
    if (numa) (void)sched_setaffinity(0, sizeof(ps->node_cpus), &ps->node_cpus);
    if (!ps->pool) {
        int want = (int)std::min<unsigned>(std::max(2u, std::thread::hardware_concurrency() / 2), 16u);
        if (const char* e = getenv("UGVC_HOST_THREADS")) want = std::max(2, atoi(e));
        ps->pool = new HostPool(want - 1, numa ? &ps->node_cpus : nullptr);
    }
    if (!ps->h2d) UGVC_HIP(hipStreamCreateWithFlags(&ps->h2d, hipStreamNonBlocking));
    if (!ps->d2h) UGVC_HIP(hipStreamCreateWithFlags(&ps->d2h, hipStreamNonBlocking));
    // (the two allele-offset columns sit at the END of a slot: a chunk whose pool is canonical is shipped without them)
    const Col cols[] = {{v->contig, &ctx->v_contig, 2}, {v->pos, &ctx->v_pos, 4},   {v->ref_len, &ctx->v_rl, 2}, {v->alt_len, &ctx->v_al, 2},
                        {v->qual, &ctx->v_qual, 4},     {v->sor, &ctx->v_sor, 4},   {v->dp, &ctx->v_dp, 4},      {v->ad_ref, &ctx->v_adr, 4},
                        {v->ad_alt, &ctx->v_ada, 4},    {v->gq, &ctx->v_gq, 1},     {v->ref_off, &ctx->v_ro, 4}, {v->alt_off, &ctx->v_ao, 4}};
    constexpr int NC = sizeof(cols) / sizeof(cols[0]);
    constexpr int kColRo = NC - 2, kColAo = NC - 1;
    const char* dv_env = getenv("UGVC_PIPE_DERIVE");
    const bool derive_offsets = dv_env ? atoi(dv_env) != 0 : true;
    size_t row_bytes = 0;
    for (const Col& c : cols) row_bytes += c.w;
    {
        // A resident column that must grow is freed and allocated afresh: whatever callset was resident is gone from that
        // moment on, whether this call then succeeds, fails or was only a reservation (ugvc_reserve) - the context says so
        // before the first buffer moves (a later ugvc_filter_resident / ugvc_results_download / ugvc_sec_apply
        // used to find the old row count beside uninitialised columns).
        bool grows = (size_t)v->alleles_len + 16 > ctx->v_alleles.cap || (size_t)n * 4 > ctx->r_score.cap || (size_t)n > ctx->r_filter.cap ||
                     (size_t)n > ctx->r_flags.cap;
        for (const Col& c : cols) grows |= (size_t)n * c.w > c.dst->cap;
        if (grows && ctx->n > 0) {
            UGVC_HIP(hipStreamSynchronize(ctx->stream));
            ctx->n = 0; ctx->n_indel = 0; ctx->scored = 0; ctx->density_n = 0;
        }
    }
    for (const Col& c : cols)
        if (ensure(*c.dst, (size_t)n * c.w)) return -1;
    if (ensure(ctx->v_alleles, (size_t)v->alleles_len + 16)) return -1;
    if (ensure(ctx->r_score, (size_t)n * 4) || ensure(ctx->r_filter, (size_t)n) || ensure(ctx->r_flags, (size_t)n)) return -1;
    // chunk bounds (multiples of 64 rows): equal chunks, except that the first and the last are cut again into 1/4 + 1/4 + 1/2 -
    // nothing overlaps the packing of the first chunk nor the pass + download of the last, so those two are kept short
    // (UGVC_PIPE_TAPER=0: plain equal chunks)
    std::vector<int64_t> cb;
    {
        const char* te = getenv("UGVC_PIPE_TAPER");
        const bool taper = (te ? atoi(te) != 0 : true) && n_chunks >= 4;
        const int64_t unit = ((n + n_chunks - 1) / n_chunks + 63) & ~(int64_t)63;
        const int64_t q = std::max<int64_t>(64, (unit / 4 + 63) & ~(int64_t)63);
        std::vector<int64_t> sizes;
        int64_t left = n;
        auto take = [&](int64_t m) { m = std::min(m, left); if (m > 0) { sizes.push_back(m); left -= m; } };
        if (taper) { take(q); take(q); take(unit - 2 * q); }
        while (left > (taper ? unit : 0)) take(unit);
        if (taper && left > 0) {
            // the remainder (at most one unit): 1/2 + 1/4 + 1/4 of it
            const int64_t h = (left / 2 + 63) & ~(int64_t)63, r = ((left - h) / 2 + 63) & ~(int64_t)63;
            take(h); take(r); take(left);
        }
        int64_t at = 0;
        for (int64_t m : sizes) { cb.push_back(at); at += m; }
        cb.push_back(n);
    }
    const int K = (int)cb.size() - 1;
    int64_t rows_chunk = 0;
    for (int c = 0; c < K; ++c) rows_chunk = std::max(rows_chunk, cb[(size_t)c + 1] - cb[(size_t)c]);
    // a slot = one chunk's twelve columns back to back (64-byte aligned) | its three result columns: ONE copy each way per
    // chunk.  (Twelve copies per chunk - ~2 MB pieces - cost ~1 ms per hundred in fixed overheads: tools/calib/pcie_probe.hip,
    // 3.65 ms for one 200 MB copy against 4.64 ms for 96 pieces; measured on this call: 6.2 ms with per-column copies.)
    const size_t slot_bytes = (size_t)rows_chunk * row_bytes + 64 * NC + 256, res_bytes = (size_t)rows_chunk * 6 + 64 * 3;
    for (int k = 0; k < NS; ++k) {
        if (pinned(ps->stage[k], ps->stage_cap[k], slot_bytes)) return -1;
        if (ensure(ps->d_stage[k], slot_bytes)) return -1;
        if (ensure(ps->d_sums[k], ((size_t)rows_chunk / (256 * kOffRows) + 2) * 4)) return -1;
    }
    if (pinned(ps->res, ps->res_cap, (size_t)K * res_bytes)) return -1;
    if (ensure(ps->d_res, (size_t)K * res_bytes)) return -1;
    if (pinned(ps->alle, ps->alle_cap, (size_t)v->alleles_len + 16)) return -1;
    if (trace && !ps->ev_timed) {
        for (hipEvent_t e : ps->ev) (void)hipEventDestroy(e);
        ps->ev.clear();
        ps->ev_timed = true;
        UGVC_HIP(hipEventCreate(&ps->ev_t0));
    }
    while (ps->ev.size() < (size_t)(4 * K)) {
        hipEvent_t e;
        UGVC_HIP(hipEventCreateWithFlags(&e, ps->ev_timed ? hipEventDefault : hipEventDisableTiming));
        ps->ev.push_back(e);
    }
    if (reserve_only) return 0;                                // (ugvc_reserve: buffers, pinned slots, streams, events, worker pool - no data touched)
    auto ev_in = [&](int c) { return ps->ev[(size_t)(4 * c)]; };          // the chunk's slot has landed in device staging
    auto ev_pass = [&](int c) { return ps->ev[(size_t)(4 * c + 1)]; };    // the pass over the chunk is done
    auto ev_out = [&](int c) { return ps->ev[(size_t)(4 * c + 2)]; };     // the chunk's results are in pinned memory
    auto ev_placed = [&](int c) { return ps->ev[(size_t)(4 * c + 3)]; };  // columns and results copied into the resident columns

    HostPool& pool = *ps->pool;
    const int T = pool.size();
    struct Burst {
        HostPool& p;
        explicit Burst(HostPool& q) : p(q) { p.burst(true); }
        ~Burst() { p.burst(false); }
    } burst_guard(pool);
    const char* nt_env = getenv("UGVC_PIPE_NT");
    const bool stream_stores = nt_env ? atoi(nt_env) != 0 : true;
    mark(-1, "setup");
    if (trace) UGVC_HIP(hipEventRecord(ps->ev_t0, ps->h2d));

    std::atomic<int64_t> bad_row{INT64_MAX};
    std::atomic<int> bad_what{0};
    std::vector<std::atomic<int64_t>> chunk_indel((size_t)K);
    for (auto& x : chunk_indel) x.store(0);
    std::vector<std::atomic<int>> chunk_canon((size_t)K);      // 1: the chunk's allele offsets follow from the lengths
    for (auto& x : chunk_canon) x.store(derive_offsets ? 1 : 0);
    const int n_contigs = ctx->n_contigs;
    auto slot_offsets = [&](int64_t m, size_t (&off)[NC]) {
        size_t used = 0;
        for (int q = 0; q < NC; ++q) { off[q] = used; used += ((size_t)m * cols[q].w + 63) & ~(size_t)63; }
        return used;
    };
    auto res_off = [&](int c, int64_t m, size_t (&o)[3]) {
        o[0] = (size_t)c * res_bytes;
        o[1] = o[0] + (((size_t)m * 4 + 63) & ~(size_t)63);
        o[2] = o[1] + (((size_t)m + 63) & ~(size_t)63);
    };
    // piece t of T of chunk c: the checks - the rows are what the kernels rely on; do the allele offsets follow from the
    // lengths (then they are neither copied nor shipped) - and the copy of the ten other columns into the chunk's pinned
    // slot in the same job; the two offset columns follow in a second job only where a chunk needs them
    auto copy_cols = [&](int c, int t, int q0, int q1) {
        const int64_t a = cb[(size_t)c], m = cb[(size_t)c + 1] - a;
        const int64_t lo = a + m * t / T, hi = a + m * (t + 1) / T;
        if (hi <= lo) return;
        uint8_t* st = static_cast<uint8_t*>(ps->stage[c % NS]);
        size_t off[NC];
        (void)slot_offsets(m, off);
        for (int q = q0; q < q1; ++q) {
            uint8_t* d = st + off[q] + (size_t)(lo - a) * cols[q].w;
            const uint8_t* sp = static_cast<const uint8_t*>(cols[q].src) + (size_t)lo * cols[q].w;
            if (stream_stores) copy_stream(d, sp, (size_t)(hi - lo) * cols[q].w);
            else memcpy(d, sp, (size_t)(hi - lo) * cols[q].w);
        }
        if (stream_stores) copy_stream_fence();
    };
    auto check_piece = [&](int c, int t) {
        const int64_t a = cb[(size_t)c], m = cb[(size_t)c + 1] - a;
        const int64_t lo = a + m * t / T, hi = a + m * (t + 1) / T;
        if (hi <= lo) return;
        int64_t ind = 0, row = -1;
        if (const int what = validate_rows(v, lo, hi, n_contigs, &ind, &row)) {
            int64_t cur = bad_row.load();
            while (row < cur && !bad_row.compare_exchange_weak(cur, row)) {}
            if (bad_row.load() == row) bad_what.store(what);
            return;
        }
        chunk_indel[(size_t)c].fetch_add(ind);
        if (chunk_canon[(size_t)c].load(std::memory_order_relaxed) && !offsets_canonical(v, lo, hi, a)) chunk_canon[(size_t)c].store(0);
        copy_cols(c, t, 0, kColRo);
    };
    auto pack_piece = [&](int c, int t) { copy_cols(c, t, kColRo, NC); };       // (the offset columns of a non-canonical chunk)
    // piece t of T of chunk c's results: pinned memory -> the caller's arrays
    auto out_piece = [&](int c, int t) {
        const int64_t a = cb[(size_t)c], m = cb[(size_t)c + 1] - a;
        size_t o[3];
        res_off(c, m, o);
        const uint8_t* r = static_cast<const uint8_t*>(ps->res);
        const int64_t lo = m * t / T, hi = m * (t + 1) / T;
        if (out->tree_score) memcpy(out->tree_score + a + lo, r + o[0] + (size_t)lo * 4, (size_t)(hi - lo) * 4);
        if (out->filter) memcpy(out->filter + a + lo, r + o[1] + (size_t)lo, (size_t)(hi - lo));
        if (out->flags) memcpy(out->flags + a + lo, r + o[2] + (size_t)lo, (size_t)(hi - lo));
    };
    // ---- head: the allele pool (codes 0..4, 16 zero bytes behind it: allele tails are fetched with fixed-width loads) and
    // chunk 0 are packed in one go
    {
        uint8_t* al = static_cast<uint8_t*>(ps->alle);
        const size_t len = (size_t)v->alleles_len;
        pool.parallel_for(2 * T, [&](int task) {
            if (task < T) {
                const size_t lo = len * (size_t)task / (size_t)T, hi = len * (size_t)(task + 1) / (size_t)T;
                memcpy(al + lo, v->alleles + lo, hi - lo);
            } else {
                check_piece(0, task - T);
            }
        });
        if (bad_row.load() == INT64_MAX && !chunk_canon[0].load()) pool.parallel_for(T, [&](int task) { pack_piece(0, task); });
        ::memset(al + len, 0, 16);
        mark(0, "head_packed");
        // everything queued on the context stream so far (model uploads, an earlier resident pass ...) precedes the first copy:
        // the copy stream must not overwrite the allele pool or columns such a pass may still be reading
        UGVC_HIP(hipStreamSynchronize(ctx->stream));
        UGVC_HIP(hipMemcpyAsync(ctx->v_alleles.p, al, len + 16, hipMemcpyHostToDevice, ps->h2d));
    }
    // every error exit below leaves the context EMPTY (earlier chunks have already been placed into the resident columns: a
    // later ugvc_filter_resident / ugvc_results_download must not see a mixed callset) with the three streams idle - the
    // pinned slots may still be in flight otherwise
    struct Fail {
        ugvc_ctx* c; PipeState* p; bool armed;
        ~Fail() {
            if (!armed) return;
            (void)hipStreamSynchronize(p->h2d);
            (void)hipStreamSynchronize(c->stream);
            (void)hipStreamSynchronize(p->d2h);
            c->n = 0; c->n_indel = 0; c->scored = 0; c->density_n = 0;
        }
    } fail_guard{ctx, ps, true};
    // kernel selection (32-bit offset limits of v5 / fm5 / v3) is decided on THIS callset's size, not the previous one's
    ctx->n = n;
    ctx->scored = 0;
    FilterArgs base;
    if (build_args(ctx, base, false)) return -1;
    mark(-1, "args");

    int64_t n_indel_total = 0;
    int rc = 0, issued = 0;
    int job_pack = -1, job_out = -1;
    const std::function<void(int)> job = [&](int task) {
        if (task < T) { if (job_pack >= 0) check_piece(job_pack, task); }
        else if (job_out >= 0) out_piece(job_out, task - T);
    };
    const std::function<void(int)> job2 = [&](int task) { pack_piece(job_pack, task); };
    for (int c = 0; c < K && !rc && bad_row.load() == INT64_MAX; ++c) {
        const int64_t a = cb[(size_t)c], b = cb[(size_t)c + 1], m = b - a;
        const int slot = c % NS;
        size_t off[NC];
        const size_t used = slot_offsets(m, off);
        n_indel_total += chunk_indel[(size_t)c].load();
        // ---- one DMA for the whole slot; the device block of this slot is free once chunk c - NS has been placed
        uint8_t* st = static_cast<uint8_t*>(ps->stage[slot]);
        uint8_t* dst = static_cast<uint8_t*>(ps->d_stage[slot].p);
        if (c >= NS) UGVC_HIP(hipStreamWaitEvent(ps->h2d, ev_placed(c - NS), 0));
        const bool canon = chunk_canon[(size_t)c].load() != 0;
        UGVC_HIP(hipMemcpyAsync(dst, st, canon ? off[kColRo] : used, hipMemcpyHostToDevice, ps->h2d));
        UGVC_HIP(hipEventRecord(ev_in(c), ps->h2d));
        // ---- the pass over the chunk reads its columns where the DMA put them and writes its results packed
        size_t o[3];
        res_off(c, m, o);
        uint8_t* dr = static_cast<uint8_t*>(ps->d_res.p);
        UGVC_HIP(hipStreamWaitEvent(ctx->stream, ev_in(c), 0));
        if (canon) {
            const unsigned nb = (unsigned)((m + 256 * kOffRows - 1) / (256 * kOffRows));
            uint32_t* sums = ps->d_sums[slot].as<uint32_t>();
            const uint16_t* rl_d = reinterpret_cast<const uint16_t*>(dst + off[2]);
            const uint16_t* al_d = reinterpret_cast<const uint16_t*>(dst + off[3]);
            UGVC_LAUNCH(pipe_offsets_sum_kernel, dim3(nb), dim3(256), 0, ctx->stream, rl_d, al_d, m, sums);
            UGVC_LAUNCH(pipe_offsets_fill_kernel, dim3(nb), dim3(256), 0, ctx->stream, rl_d, al_d, m, v->ref_off[a], (const uint32_t*)sums,
                               reinterpret_cast<uint32_t*>(dst + off[kColRo]), reinterpret_cast<uint32_t*>(dst + off[kColAo]));
            UGVC_HIP(hipGetLastError());
        }
        FilterArgs fa = base;
        fa.n = m;
        fa.contig = reinterpret_cast<const uint16_t*>(dst + off[0]);
        fa.pos = reinterpret_cast<const int32_t*>(dst + off[1]);
        fa.ref_len = reinterpret_cast<const uint16_t*>(dst + off[2]);
        fa.alt_len = reinterpret_cast<const uint16_t*>(dst + off[3]);
        fa.qual = reinterpret_cast<const float*>(dst + off[4]);
        fa.sor = reinterpret_cast<const float*>(dst + off[5]);
        fa.dp = reinterpret_cast<const int32_t*>(dst + off[6]);
        fa.ad_ref = reinterpret_cast<const int32_t*>(dst + off[7]);
        fa.ad_alt = reinterpret_cast<const int32_t*>(dst + off[8]);
        fa.gq = reinterpret_cast<const uint8_t*>(dst + off[9]);
        fa.ref_off = reinterpret_cast<const uint32_t*>(dst + off[kColRo]);
        fa.alt_off = reinterpret_cast<const uint32_t*>(dst + off[kColAo]);
        fa.score = reinterpret_cast<float*>(dr + o[0]);
        fa.filter = dr + o[1];
        fa.flags = dr + o[2];
        // (sizes the indel tiles' table slices: the callset-wide count, estimated from this chunk's share - chunks differ in size)
        ctx->n_indel = (int64_t)((double)chunk_indel[(size_t)c].load() * (double)n / (double)m);
        ctx->density_n = n;                                 // (table rows per tile are a property of the whole callset)
        rc = launch_score(ctx, fa);
        ctx->density_n = 0;
        if (rc) break;
        UGVC_HIP(hipEventRecord(ev_pass(c), ctx->stream));
        UGVC_HIP(hipStreamWaitEvent(ps->d2h, ev_pass(c), 0));
        UGVC_HIP(hipMemcpyAsync(static_cast<uint8_t*>(ps->res) + o[0], dr + o[0], o[2] + (size_t)m - o[0], hipMemcpyDeviceToHost, ps->d2h));
        UGVC_HIP(hipEventRecord(ev_out(c), ps->d2h));
        // ---- behind the pass: columns and results into their resident places (what ugvc_filter_resident / ugvc_results_download
        // / ugvc_feature_matrix find afterwards)
        {
            SegTable t;
            for (int q = 0; q < NC; ++q)
                t.s[q] = Seg{dst + off[q], static_cast<uint8_t*>(cols[q].dst->p) + (size_t)a * cols[q].w, (uint64_t)m * cols[q].w};
            t.s[NC] = Seg{dr + o[0], reinterpret_cast<uint8_t*>(ctx->r_score.as<float>() + a), (uint64_t)m * 4};
            t.s[NC + 1] = Seg{dr + o[1], ctx->r_filter.as<uint8_t>() + a, (uint64_t)m};
            t.s[NC + 2] = Seg{dr + o[2], ctx->r_flags.as<uint8_t>() + a, (uint64_t)m};
            UGVC_LAUNCH(pipe_place_kernel, dim3(64, NC + 3), dim3(256), 0, ctx->stream, t);
            UGVC_HIP(hipGetLastError());
            UGVC_HIP(hipEventRecord(ev_placed(c), ctx->stream));
        }
        issued = c + 1;
        mark(c, "issued");
        // ---- while the chunk is in flight the pool packs the next one and hands an earlier one's results to the caller
        job_pack = c + 1 < K ? c + 1 : -1;
        job_out = c >= 2 ? c - 2 : -1;
        if (job_pack >= NS) UGVC_HIP(hipEventSynchronize(ev_in(job_pack - NS)));   // the DMA out of that pinned slot has finished
        if (job_out >= 0) UGVC_HIP(hipEventSynchronize(ev_out(job_out)));
        mark(c, "job_start");
        if (job_pack >= 0 || job_out >= 0) pool.parallel_for(2 * T, job);
        if (job_pack >= 0 && bad_row.load() == INT64_MAX && !chunk_canon[(size_t)job_pack].load()) pool.parallel_for(T, job2);
        mark(c, "job_end");
    }
    if (bad_row.load() != INT64_MAX || rc) {
        (void)hipStreamSynchronize(ps->h2d);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamSynchronize(ps->d2h);
        if (rc) return -1;
        // (name the row again on this thread: two pieces may have reported different rows at the same time)
        int64_t i = bad_row.load(), ind = 0, row = i;
        const int what = validate_rows(v, i, i + 1, n_contigs, &ind, &row);
        return fail(std::string(row_error_text(what ? what : bad_what.load())) + std::to_string(i));
    }
    mark(K, "loop_end");
    for (int c = std::max(0, issued - 2); c < issued; ++c) {
        UGVC_HIP(hipEventSynchronize(ev_out(c)));
        mark(c, "out_copy");
        pool.parallel_for(T, [&](int t) { out_piece(c, t); });
    }
    UGVC_HIP(hipStreamSynchronize(ps->h2d));
    UGVC_HIP(hipStreamSynchronize(ctx->stream));
    mark(K, "done");
    if (trace) {
        for (const HostMark& m : marks) fprintf(stderr, "[pipe] host chunk %d %s %.3f\n", m.chunk, m.what, m.ms);
        const char* names[4] = {"h2d_done", "pass_done", "d2h_done", "placed"};
        for (int c = 0; c < K; ++c)
            for (int q = 0; q < 4; ++q) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, ps->ev_t0, ps->ev[(size_t)(4 * c + q)]) == hipSuccess)
                    fprintf(stderr, "[pipe] dev chunk %d %s %.3f\n", c, names[q], ms);
            }
    }
    fail_guard.armed = false;
    ctx->n = n;
    ctx->n_indel = n_indel_total;
    ctx->scored = 1;
    return 0;
}

}  // namespace ugvc
