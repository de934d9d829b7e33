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
#include <string.h>

#include <atomic>
#include <fstream>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include "ugvc_v2.hpp"

namespace ugvc {

int launch_score(ugvc_ctx* ctx, const FilterArgs& a);

// ---- a persistent pool: parallel_for(n_tasks, f) runs f(task) on the workers and on the caller
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
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    void parallel_for(int n_tasks, const std::function<void(int)>& f) {
        if (n_tasks <= 0) return;
        {
            std::lock_guard<std::mutex> g(m_);
            job_ = &f;
            n_tasks_ = n_tasks;
            next_.store(0);
            pending_ = n_tasks;
            ++gen_;
        }
        cv_.notify_all();
        drain();                                             // the caller works too
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return pending_ == 0; });
        job_ = nullptr;
    }
    // the same without the caller: start() returns at once, wait() blocks until the tasks are done
    void start(int n_tasks, const std::function<void(int)>& f) {
        if (n_tasks <= 0) return;
        {
            std::lock_guard<std::mutex> g(m_);
            job_ = &f;
            n_tasks_ = n_tasks;
            next_.store(0);
            pending_ = n_tasks;
            ++gen_;
        }
        cv_.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return pending_ == 0; });
        job_ = nullptr;
    }
    int size() const { return (int)th_.size() + 1; }
    int workers() const { return (int)th_.size(); }

  private:
    void drain() {
        for (;;) {
            const int t = next_.fetch_add(1);
            if (t >= n_tasks_) return;
            (*job_)(t);
            std::lock_guard<std::mutex> g(m_);
            if (--pending_ == 0) done_.notify_all();
        }
    }
    void work() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return stop_ || (gen_ != seen && job_ != nullptr); });
                if (stop_) return;
                seen = gen_;
            }
            drain();
        }
    }
    cpu_set_t cpus_;
    bool pinned_ = false;
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(int)>* job_ = nullptr;
    std::atomic<int> next_{0};
    int n_tasks_ = 0, pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

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
    HostPool* pool = nullptr;
    hipStream_t h2d = nullptr, d2h = nullptr;
    void* stage[2] = {nullptr, nullptr};        // pinned: one chunk's columns, back to back
    size_t stage_cap[2] = {0, 0};
    DeviceBuf d_stage[2];                       // where a slot lands on the device before it is scattered into the columns
    void* res = nullptr;                        // pinned: every chunk's packed result columns
    size_t res_cap = 0;
    DeviceBuf d_res;
    void* alle = nullptr;                       // pinned: the allele pool
    size_t alle_cap = 0;
    std::vector<hipEvent_t> ev;                 // four per chunk
};

static PipeState* pipe_state(ugvc_ctx* ctx) {
    if (!ctx->pipe) ctx->pipe = new PipeState();
    return static_cast<PipeState*>(ctx->pipe);
}

void pipe_destroy(ugvc_ctx* ctx) {
    if (!ctx->pipe) return;
    PipeState* p = static_cast<PipeState*>(ctx->pipe);
    delete p->pool;
    for (void* q : {p->stage[0], p->stage[1], p->res, p->alle})
        if (q) (void)hipHostFree(q);
    for (DeviceBuf* b : {&p->d_stage[0], &p->d_stage[1], &p->d_res})
        if (b->p) (void)hipFree(b->p);
    for (hipEvent_t e : p->ev) (void)hipEventDestroy(e);
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

int filter_variants_pipelined(ugvc_ctx* ctx, const ugvc_variants* v, const ugvc_results* out, int n_chunks) {
    const int64_t n = v->n;
    PipeState* ps = pipe_state(ctx);
    // pool and pinned staging live on the GPU's NUMA node: the calling thread is moved there for the duration of the call
    // (it packs and hands over too) and put back on its own CPUs afterwards
    cpu_set_t node_cpus, mine;
    const bool numa = gpu_node_cpus(ctx->device, node_cpus) && sched_getaffinity(0, sizeof(mine), &mine) == 0;
    struct Restore {
        bool on; cpu_set_t set;
        ~Restore() { if (on) (void)sched_setaffinity(0, sizeof(set), &set); }
    } restore{numa, mine};
    if (numa) (void)sched_setaffinity(0, sizeof(node_cpus), &node_cpus);
    if (!ps->pool) {
        int want = (int)std::min<unsigned>(std::max(2u, std::thread::hardware_concurrency() / 2), 16u);
        if (const char* e = getenv("UGVC_HOST_THREADS")) want = std::max(1, atoi(e));
        ps->pool = new HostPool(want - 1, numa ? &node_cpus : nullptr);
    }
    if (!ps->h2d) UGVC_HIP(hipStreamCreateWithFlags(&ps->h2d, hipStreamNonBlocking));
    if (!ps->d2h) UGVC_HIP(hipStreamCreateWithFlags(&ps->d2h, hipStreamNonBlocking));
    const Col cols[] = {{v->contig, &ctx->v_contig, 2}, {v->pos, &ctx->v_pos, 4},   {v->ref_len, &ctx->v_rl, 2}, {v->alt_len, &ctx->v_al, 2},
                        {v->ref_off, &ctx->v_ro, 4},    {v->alt_off, &ctx->v_ao, 4}, {v->qual, &ctx->v_qual, 4},  {v->sor, &ctx->v_sor, 4},
                        {v->dp, &ctx->v_dp, 4},         {v->ad_ref, &ctx->v_adr, 4}, {v->ad_alt, &ctx->v_ada, 4}, {v->gq, &ctx->v_gq, 1}};
    constexpr int NC = sizeof(cols) / sizeof(cols[0]);
    size_t row_bytes = 0;
    for (const Col& c : cols) row_bytes += c.w;
    for (const Col& c : cols)
        if (ensure(*c.dst, (size_t)n * c.w)) return -1;
    if (ensure(ctx->v_alleles, (size_t)v->alleles_len + 16)) return -1;
    if (ensure(ctx->r_score, (size_t)n * 4) || ensure(ctx->r_filter, (size_t)n) || ensure(ctx->r_flags, (size_t)n)) return -1;
    // chunk bounds: equal chunks (half-size first and last chunks - nothing overlaps the staging of the first nor the tail of
    // the last - measured no better: 5.6 against 5.3 ms)
    std::vector<int64_t> cb;
    {
        const int64_t unit = ((n + n_chunks - 1) / n_chunks + 63) & ~(int64_t)63;
        for (int64_t at = 0; at < n; at += unit) cb.push_back(at);
        cb.push_back(n);
    }
    const int K = (int)cb.size() - 1;
    int64_t rows_chunk = 0;
    for (int c = 0; c < K; ++c) rows_chunk = std::max(rows_chunk, cb[(size_t)c + 1] - cb[(size_t)c]);
    // a slot = one chunk's twelve columns back to back (64-byte aligned) | its three result columns: ONE copy each way per
    // chunk.  (Twelve copies per chunk - ~2 MB pieces - cost ~1 ms per hundred in fixed overheads: tools/calib/pcie_probe.hip,
    // 3.65 ms for one 200 MB copy against 4.64 ms for 96 pieces; measured on this call: 6.2 ms with per-column copies.)
    const size_t slot_bytes = (size_t)rows_chunk * row_bytes + 64 * NC, res_bytes = (size_t)rows_chunk * 6 + 64 * 3;
    for (int k = 0; k < 2; ++k) {
        if (pinned(ps->stage[k], ps->stage_cap[k], slot_bytes)) return -1;
        if (ensure(ps->d_stage[k], slot_bytes)) return -1;
    }
    if (pinned(ps->res, ps->res_cap, (size_t)K * res_bytes)) return -1;
    if (ensure(ps->d_res, (size_t)K * res_bytes)) return -1;
    if (pinned(ps->alle, ps->alle_cap, (size_t)v->alleles_len + 16)) return -1;
    while (ps->ev.size() < (size_t)(4 * K)) {
        hipEvent_t e;
        UGVC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ps->ev.push_back(e);
    }
    auto ev_in = [&](int c) { return ps->ev[(size_t)(4 * c)]; };          // the chunk's slot has landed in device staging
    auto ev_scat = [&](int c) { return ps->ev[(size_t)(4 * c + 1)]; };    // ... and has been scattered into the resident columns
    auto ev_pass = [&](int c) { return ps->ev[(size_t)(4 * c + 2)]; };    // the pass over the chunk is done, results packed
    auto ev_out = [&](int c) { return ps->ev[(size_t)(4 * c + 3)]; };     // the chunk's results are in pinned memory

    HostPool& pool = *ps->pool;
    const int T = pool.size();
    // ---- the allele pool first (codes 0..4, 16 zero bytes behind it: allele tails are fetched with fixed-width loads)
    {
        uint8_t* al = static_cast<uint8_t*>(ps->alle);
        const size_t len = (size_t)v->alleles_len;
        pool.parallel_for(T, [&](int t) {
            const size_t lo = len * (size_t)t / (size_t)T, hi = len * (size_t)(t + 1) / (size_t)T;
            memcpy(al + lo, v->alleles + lo, hi - lo);
        });
        ::memset(al + len, 0, 16);
        UGVC_HIP(hipMemcpyAsync(ctx->v_alleles.p, al, len + 16, hipMemcpyHostToDevice, ps->h2d));
    }
    // everything queued on the context stream so far (model uploads ...) precedes the first pass; the copy stream must
    // not overwrite columns an earlier pass may still be reading
    UGVC_HIP(hipStreamSynchronize(ctx->stream));

    FilterArgs base;
    if (build_args(ctx, base, false)) return -1;
    std::atomic<int64_t> bad_row{INT64_MAX};
    std::atomic<int> bad_what{0};
    int64_t n_indel_total = 0;
    const int n_contigs = ctx->n_contigs;
    int rc = 0;
    int copied_out = 0;                                    // chunks whose results have reached the caller's arrays
    auto res_off = [&](int c, int64_t m, size_t (&o)[3]) {
        o[0] = (size_t)c * res_bytes;
        o[1] = o[0] + (((size_t)m * 4 + 63) & ~(size_t)63);
        o[2] = o[1] + (((size_t)m + 63) & ~(size_t)63);
    };
    auto copy_out = [&](int c) -> int {
        UGVC_HIP(hipEventSynchronize(ev_out(c)));
        const int64_t a = cb[(size_t)c], m = cb[(size_t)c + 1] - a;
        size_t o[3];
        res_off(c, m, o);
        const uint8_t* r = static_cast<const uint8_t*>(ps->res);
        pool.parallel_for(T, [&](int t) {
            const int64_t lo = m * t / T, hi = m * (t + 1) / T;
            if (out->tree_score) memcpy(out->tree_score + a + lo, r + o[0] + (size_t)lo * 4, (size_t)(hi - lo) * 4);
            if (out->filter) memcpy(out->filter + a + lo, r + o[1] + (size_t)lo, (size_t)(hi - lo));
            if (out->flags) memcpy(out->flags + a + lo, r + o[2] + (size_t)lo, (size_t)(hi - lo));
        });
        return 0;
    };
    for (int c = 0; c < K && !rc; ++c) {
        const int64_t a = cb[(size_t)c], b = cb[(size_t)c + 1], m = b - a;
        const int slot = c & 1;
        if (c >= 2) UGVC_HIP(hipEventSynchronize(ev_in(c - 2)));        // the DMA out of this pinned slot (chunk c - 2) has finished
        uint8_t* st = static_cast<uint8_t*>(ps->stage[slot]);
        size_t off[NC], used = 0;
        for (int q = 0; q < NC; ++q) { off[q] = used; used += ((size_t)m * cols[q].w + 63) & ~(size_t)63; }
        std::atomic<int64_t> n_indel{0};
        pool.parallel_for(T, [&](int t) {
            const int64_t lo = a + m * t / T, hi = a + m * (t + 1) / T;
            if (hi <= lo) return;
            // validation of rows [lo, hi) (what the kernels rely on), then the copy
            int64_t ind = 0;
            for (int64_t i = lo; i < hi; ++i) {
                int what = 0;
                ind += v->ref_len[i] != v->alt_len[i] ? 1 : 0;
                if (v->contig[i] >= n_contigs) what = 1;
                else if (v->ref_len[i] == 0 || v->alt_len[i] == 0) what = 2;
                else if ((int64_t)v->ref_off[i] + v->ref_len[i] > v->alleles_len || (int64_t)v->alt_off[i] + v->alt_len[i] > v->alleles_len) what = 3;
                else if (v->pos[i] < 1) what = 4;
                else if (i && (v->contig[i] < v->contig[i - 1] || (v->contig[i] == v->contig[i - 1] && v->pos[i] < v->pos[i - 1]))) what = 5;
                if (what) {
                    int64_t cur = bad_row.load();
                    while (i < cur && !bad_row.compare_exchange_weak(cur, i)) {}
                    if (bad_row.load() == i) bad_what.store(what);
                    break;
                }
            }
            n_indel.fetch_add(ind);
            for (int q = 0; q < NC; ++q)
                memcpy(st + off[q] + (size_t)(lo - a) * cols[q].w, static_cast<const uint8_t*>(cols[q].src) + (size_t)lo * cols[q].w,
                       (size_t)(hi - lo) * cols[q].w);
        });
        if (bad_row.load() != INT64_MAX) break;
        n_indel_total += n_indel.load();
        // ---- one DMA for the whole slot; the device copy of slot (c & 1) is free once chunk c - 2 has been scattered
        uint8_t* dst = static_cast<uint8_t*>(ps->d_stage[slot].p);
        if (c >= 2) UGVC_HIP(hipStreamWaitEvent(ps->h2d, ev_scat(c - 2), 0));
        UGVC_HIP(hipMemcpyAsync(dst, st, used, hipMemcpyHostToDevice, ps->h2d));
        UGVC_HIP(hipEventRecord(ev_in(c), ps->h2d));
        // ---- on the context stream: scatter into the resident columns (device copies at HBM rate), then the pass over rows [a, b)
        UGVC_HIP(hipStreamWaitEvent(ctx->stream, ev_in(c), 0));
        for (int q = 0; q < NC; ++q)
            UGVC_HIP(hipMemcpyAsync(static_cast<uint8_t*>(cols[q].dst->p) + (size_t)a * cols[q].w, dst + off[q], (size_t)m * cols[q].w,
                                    hipMemcpyDeviceToDevice, ctx->stream));
        UGVC_HIP(hipEventRecord(ev_scat(c), ctx->stream));
        FilterArgs fa = base;
        fa.n = m;
        fa.contig += a; fa.pos += a; fa.ref_len += a; fa.alt_len += a; fa.ref_off += a; fa.alt_off += a;
        fa.qual += a; fa.sor += a; fa.dp += a; fa.ad_ref += a; fa.ad_alt += a; fa.gq += a;
        fa.score += a; fa.filter += a; fa.flags += a;
        ctx->n_indel = n_indel.load() * K;                  // (sizes the indel tiles' table slices: the callset-wide rate, from this chunk)
        ctx->density_n = n;                                 // (table rows per tile are a property of the whole callset)
        rc = launch_score(ctx, fa);
        ctx->density_n = 0;
        if (rc) break;
        // ---- the chunk's three result columns, packed, in one copy
        size_t o[3];
        res_off(c, m, o);
        uint8_t* dr = static_cast<uint8_t*>(ps->d_res.p);
        UGVC_HIP(hipMemcpyAsync(dr + o[0], ctx->r_score.as<float>() + a, (size_t)m * 4, hipMemcpyDeviceToDevice, ctx->stream));
        UGVC_HIP(hipMemcpyAsync(dr + o[1], ctx->r_filter.as<uint8_t>() + a, (size_t)m, hipMemcpyDeviceToDevice, ctx->stream));
        UGVC_HIP(hipMemcpyAsync(dr + o[2], ctx->r_flags.as<uint8_t>() + a, (size_t)m, hipMemcpyDeviceToDevice, ctx->stream));
        UGVC_HIP(hipEventRecord(ev_pass(c), ctx->stream));
        UGVC_HIP(hipStreamWaitEvent(ps->d2h, ev_pass(c), 0));
        UGVC_HIP(hipMemcpyAsync(static_cast<uint8_t*>(ps->res) + o[0], dr + o[0], o[2] + (size_t)m - o[0], hipMemcpyDeviceToHost, ps->d2h));
        UGVC_HIP(hipEventRecord(ev_out(c), ps->d2h));
        // results of a chunk two behind are long back: hand them to the caller while this one is in flight
        if (c >= 2 && copy_out(copied_out++)) return -1;
    }
    if (bad_row.load() != INT64_MAX || rc) {
        (void)hipStreamSynchronize(ps->h2d);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamSynchronize(ps->d2h);
        if (rc) return -1;
        ctx->n = 0;
        ctx->scored = 0;
        const int64_t i = bad_row.load();
        switch (bad_what.load()) {
            case 1: return fail("contig index out of range at row " + std::to_string(i));
            case 2: return fail("empty allele at row " + std::to_string(i));
            case 3: return fail("allele offset outside the pool at row " + std::to_string(i));
            case 4: return fail("POS must be >= 1 at row " + std::to_string(i));
            default: return fail("variants must be sorted by (contig, pos); row " + std::to_string(i));
        }
    }
    while (copied_out < K)
        if (copy_out(copied_out++)) return -1;
    UGVC_HIP(hipStreamSynchronize(ps->h2d));
    UGVC_HIP(hipStreamSynchronize(ctx->stream));
    ctx->n = n;
    ctx->n_indel = n_indel_total;
    ctx->scored = 1;
    return 0;
}

}  // namespace ugvc
