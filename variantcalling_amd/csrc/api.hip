// C ABI of libugvc_mi355x.so: context, resident tables, hot-path entry points.
// See include/ugvc_mi355x.h for the contract and the reference interfaces each call replaces.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <limits>

#include "ugvc_prims.hpp"
#include "ugvc_v2.hpp"

namespace ugvc {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int fail(const std::string& msg) {
    g_err = msg;
    return -1;
}

int ensure(DeviceBuf& b, size_t bytes) {
    if (bytes <= b.cap && b.p) return 0;
    dev_free(b.p);
    b.p = nullptr;
    b.cap = 0;
    // (a table of a few bytes still gets 256 in production; under UGVC_GUARD it gets what was asked for, so that a read past a
    // three-entry contig_ptr or a one-key blacklist faults as well)
    size_t want = guard_on() ? std::max<size_t>(bytes, 1) : std::max<size_t>(bytes, 256);
    if (dev_alloc(&b.p, want)) { b.p = nullptr; return -1; }
    b.cap = want;
    return 0;
}

int upload(ugvc_ctx* ctx, DeviceBuf& b, const void* src, size_t bytes) {
    // every uploaded table holds at least 16 bytes, zero where nothing was uploaded: the slice loads of the scoring pass clamp
    // their row index into [0, max(rows - 1, 0)], i.e. they read row 0 of an EMPTY interval table / blacklist (the row is
    // discarded: its contig's row range is empty).  Found by the byte-exact guard mode (UGVC_GUARD_ALIGN=1) on the CLI test
    // with an empty runs file - the 256-byte floor of production allocations had been hiding it (round 5).
    if (ensure(b, std::max<size_t>(bytes, 16))) return -1;
    if (bytes < 16) UGVC_HIP(hipMemsetAsync(b.p, 0, 16, ctx->stream));
    if (bytes) UGVC_HIP(copy_in(ctx, b.p, src, bytes));
    return 0;
}

// every 64th element of a sorted table: the L2-resident first level of K0's bracket search
template <class T>
static int upload_coarse(ugvc_ctx* ctx, DeviceBuf& dst, const T* a, int64_t n) {
    std::vector<T> c((size_t)((n + 63) / 64));
    for (size_t k = 0; k < c.size(); ++k) c[k] = a[k * 64];
    return upload(ctx, dst, c.data(), c.size() * sizeof(T));
}

static void release(DeviceBuf& b) {
    if (b.p) dev_free(b.p);
    b.p = nullptr;
    b.cap = 0;
}

int build_args(ugvc_ctx* ctx, FilterArgs& a, bool want_x) {
    if (!ctx->ref.p) return fail("no reference uploaded (ugvc_ref_upload)");
    memset(&a, 0, sizeof(a));
    a.n = ctx->n;
    a.contig = ctx->v_contig.as<uint16_t>();
    a.pos = ctx->v_pos.as<int32_t>();
    a.ref_len = ctx->v_rl.as<uint16_t>();
    a.alt_len = ctx->v_al.as<uint16_t>();
    a.ref_off = ctx->v_ro.as<uint32_t>();
    a.alt_off = ctx->v_ao.as<uint32_t>();
    a.alleles = ctx->v_alleles.as<uint8_t>();
    a.qual = ctx->v_qual.as<float>();
    a.sor = ctx->v_sor.as<float>();
    a.dp = ctx->v_dp.as<int32_t>();
    a.ad_ref = ctx->v_adr.as<int32_t>();
    a.ad_alt = ctx->v_ada.as<int32_t>();
    a.gq = ctx->v_gq.as<uint8_t>();
    a.ref = ctx->ref.as<uint8_t>() + kRefFrontPad;
    a.contig_off = ctx->contig_off.as<int64_t>();
    a.runs = TrackView{ctx->runs_s.as<int32_t>(), ctx->runs_e.as<int32_t>(), ctx->runs_p.as<int32_t>(), ctx->runs_c.as<int32_t>()};
    a.has_runs = ctx->has_runs;
    a.hpol_dist = ctx->hpol_dist;
    a.mark_hpol = ctx->mark_hpol;
    a.n_tracks = ctx->n_tracks;
    for (int t = 0; t < ctx->n_tracks; ++t) {
        if (!ctx->track_set[t]) return fail("annotation track " + std::to_string(t) + " not uploaded");
        a.tracks[t] = TrackView{ctx->trk_s[t].as<int32_t>(), ctx->trk_e[t].as<int32_t>(), ctx->trk_p[t].as<int32_t>(), ctx->trk_c[t].as<int32_t>()};
    }
    a.bl = ctx->bl.as<uint64_t>();
    a.bl_coarse = ctx->bl_c.as<uint64_t>();
    a.n_bl = ctx->n_bl;
    const int F = UGVC_N_BASE_FEATURES + ctx->n_tracks;
    for (int g = 0; g < UGVC_N_GROUPS; ++g) {
        const auto& m = ctx->model[g];
        ForestView& f = a.forest[g];
        if (m.set) {
            if (m.n_features > F)
                return fail("model for group " + std::to_string(g) + " wants " + std::to_string(m.n_features) +
                            " features, engine provides " + std::to_string(F));
            f.nodes = m.nodes.as<Node>();
            f.roots = m.roots.as<int>();
            f.leaves = m.leaves.as<double2>();
            f.n_trees = m.n_trees;
            f.depth = m.depth;
            f.kind = m.kind;
            f.base = m.base;
            f.dense = m.has_dense ? m.dense.as<float2>() : nullptr;
            f.dense_leaves = m.has_dense ? m.dense_leaves.as<double2>() : nullptr;
        }
    }
    memcpy(a.flow, ctx->flow, 4);
    a.score = ctx->r_score.as<float>();
    a.filter = ctx->r_filter.as<uint8_t>();
    a.flags = ctx->r_flags.as<uint8_t>();
    a.X = want_x ? ctx->x_mat.as<float>() : nullptr;
    a.group = want_x ? ctx->x_group.as<uint8_t>() : nullptr;
    a.ablate = ctx->kernel_variant;
    a.n_contigs = ctx->n_contigs;
    return 0;
}

// The scoring pass.  v5 (kernels_v5.hip: index lists by variant class, per-wave featurize fused with the
// LDS-resident SNP forest walk, raw 16-bit codes) whenever every uploaded model is a random forest in the
// single-sum layout and the side tables are sorted / disjoint; else v3 (K0 brackets + K1 featurize / quantise +
// K2 LDS forest: pair-sum forests, XGBoost-style ensembles) when the models pack into its LDS layout; else the
// universal v1 fused kernel (any depth / threshold count / overlapping runs).
// kernel_variant bit 8 (256) forces v1, bit 16 (65536) forces v3 over v5 (A/B measurements, parity cross-checks).
int launch_score(ugvc_ctx* ctx, const FilterArgs& a) {
    if (a.flags == ctx->r_flags.as<uint8_t>()) ctx->scored = 1;
    if (!(ctx->kernel_variant & 256) && v2_available(ctx)) {
        if (!(ctx->kernel_variant & 65536) && v5_available(ctx)) return launch_filter_v5(ctx, a);
        if (v3_available(ctx)) return launch_filter_v3(ctx, a);
    }
    return launch_filter(ctx, a, true, false);
}

int validate_rows(const ugvc_variants* v, int64_t lo, int64_t hi, int n_contigs, int64_t* n_indel, int64_t* row);   // host_rows.cpp
const char* row_error_text(int what);
void pipe_destroy(ugvc_ctx* ctx);                                                          // pipeline.hip
int filter_variants_pipelined(ugvc_ctx* ctx, const ugvc_variants* v, const ugvc_results* out, int n_chunks, bool reserve_only = false);
int v5_warm(ugvc_ctx* ctx);

}  // namespace ugvc

using namespace ugvc;

extern "C" {

int ugvc_abi_version(void) { return UGVC_ABI_VERSION; }
const char* ugvc_last_error(void) { return g_err.c_str(); }

int ugvc_ctx_create(int device_id, ugvc_ctx** out) {
    if (!out) return fail("out is NULL");
    int n_dev = 0;
    UGVC_HIP(hipGetDeviceCount(&n_dev));
    if (device_id < 0 || device_id >= n_dev)
        return fail("device " + std::to_string(device_id) + " not present (" + std::to_string(n_dev) + " visible)");
    UGVC_HIP(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    UGVC_HIP(hipGetDeviceProperties(&prop, device_id));
    ugvc_ctx* ctx = new ugvc_ctx();
    ctx->device = device_id;
    ctx->n_cus = prop.multiProcessorCount;
    UGVC_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    UGVC_HIP(hipEventCreate(&ctx->ev0));
    UGVC_HIP(hipEventCreate(&ctx->ev1));
    *out = ctx;
    return 0;
}

int ugvc_comm_destroy(ugvc_ctx* ctx);

int ugvc_ctx_destroy(ugvc_ctx* ctx) {
    if (!ctx) return 0;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ugvc_comm_destroy(ctx);
    pipe_destroy(ctx);
    v2_destroy(ctx);
    bounce_destroy(ctx);
    DeviceBuf* all[] = {&ctx->ref, &ctx->contig_off, &ctx->runs_s, &ctx->runs_e, &ctx->runs_p, &ctx->runs_c, &ctx->bl, &ctx->bl_c,
                        &ctx->v_contig, &ctx->v_pos, &ctx->v_rl, &ctx->v_al, &ctx->v_ro, &ctx->v_ao,
                        &ctx->v_alleles, &ctx->v_qual, &ctx->v_sor, &ctx->v_dp, &ctx->v_adr, &ctx->v_ada,
                        &ctx->v_gq, &ctx->r_score, &ctx->r_filter, &ctx->r_flags, &ctx->x_mat, &ctx->x_group,
                        &ctx->pl_off, &ctx->pl_off32, &ctx->pl_obsb, &ctx->pl_out, &ctx->sec_keys, &ctx->sec_coarse, &ctx->sec_exp, &ctx->sec_lgtab, &ctx->g_score[0], &ctx->g_filter[0], &ctx->g_flags[0],
                        &ctx->g_score[1], &ctx->g_filter[1], &ctx->g_flags[1], &ctx->wclk_buf};
    for (auto* b : all) release(*b);
    for (int t = 0; t < UGVC_MAX_TRACKS; ++t) {
        release(ctx->trk_s[t]);
        release(ctx->trk_e[t]);
        release(ctx->trk_p[t]);
        release(ctx->trk_c[t]);
    }
    for (auto& m : ctx->model) {
        release(m.nodes); release(m.roots); release(m.leaves); release(m.dense); release(m.dense_leaves);
    }
    (void)hipEventDestroy(ctx->ev0);
    (void)hipEventDestroy(ctx->ev1);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return 0;
}

int ugvc_device_attr(ugvc_ctx* ctx, int what, int64_t* out) {
    if (!ctx || !out) return fail("NULL argument");
    hipDeviceProp_t prop;
    UGVC_HIP(hipGetDeviceProperties(&prop, ctx->device));
    switch (what) {
        case 0: *out = prop.clockRate; break;
        case 1: *out = prop.multiProcessorCount; break;
        case 2: *out = prop.memoryClockRate; break;
        case 3: *out = (int64_t)prop.sharedMemPerBlock; break;
        default: return fail("ugvc_device_attr: unknown attribute " + std::to_string(what));
    }
    return 0;
}

int ugvc_device_info(ugvc_ctx* ctx, char* name, int name_cap, int* n_cus, int64_t* hbm_bytes) {
    if (!ctx) return fail("ctx is NULL");
    hipDeviceProp_t prop;
    UGVC_HIP(hipGetDeviceProperties(&prop, ctx->device));
    if (name && name_cap > 0) {
        std::string s = std::string(prop.name) + " (" + prop.gcnArchName + ")";
        strncpy(name, s.c_str(), name_cap - 1);
        name[name_cap - 1] = 0;
    }
    if (n_cus) *n_cus = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return 0;
}

int ugvc_sync(ugvc_ctx* ctx) {
    if (!ctx) return fail("ctx is NULL");
    UGVC_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int ugvc_selftest(ugvc_ctx* ctx, int64_t n) {
    // The device canary: a copy round trip and one small kernel of this library whose answer the host knows.  A box whose
    // GPU is unusable fails HERE, with the step named, not inside the first scoring pass.
    if (!ctx) return fail("ctx is NULL");
    if (n < 1 || n > (1 << 24)) return fail("selftest size out of range");
    UGVC_HIP(hipSetDevice(ctx->device));
    std::vector<uint64_t> h((size_t)n), back((size_t)n, ~0ull);
    for (int64_t i = 0; i < n; ++i) h[(size_t)i] = (uint64_t)((i * 2654435761ll) % 97);
    DeviceBuf d, tmp, db;
    int rc = 0;
    do {
        // a byte buffer with an odd tail first (8 n + n % 7 bytes): the pinned-slot copies split large pieces over threads
        const size_t nb = (size_t)n * 8 + (size_t)(n % 7);
        std::vector<uint8_t> hb(nb), bb(nb, 0xEE);
        for (size_t i = 0; i < nb; ++i) hb[i] = (uint8_t)((i * 131u + (i >> 9)) & 0xFF);
        if ((rc = upload(ctx, db, hb.data(), nb))) break;
        if (copy_out(ctx, bb.data(), db.p, nb) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = fail("selftest: byte copy round trip failed (device error)"); break; }
        if (memcmp(hb.data(), bb.data(), nb)) {
            size_t at = 0;
            while (at < nb && hb[at] == bb[at]) ++at;
            rc = fail("selftest: byte copy round trip returned different bytes at offset " + std::to_string(at) + " of " + std::to_string(nb));
            break;
        }
        if ((rc = upload(ctx, d, h.data(), (size_t)n * 8))) break;
        if (copy_out(ctx, back.data(), d.p, (size_t)n * 8) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = fail("selftest: copy round trip failed (device error)"); break; }
        if (memcmp(h.data(), back.data(), (size_t)n * 8)) { rc = fail("selftest: copy round trip returned different bytes"); break; }
        if ((rc = scan_u64(ctx, tmp, d.as<uint64_t>(), n, true))) break;
        if (copy_out(ctx, back.data(), d.p, (size_t)n * 8) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = fail("selftest: scan kernel failed (device error)"); break; }
        uint64_t run = 0;
        for (int64_t i = 0; i < n && !rc; ++i) {
            run += h[(size_t)i];
            if (back[(size_t)i] != run) rc = fail("selftest: scan kernel wrong at element " + std::to_string(i));
        }
    } while (0);
    dev_free(d.p);
    dev_free(tmp.p);
    dev_free(db.p);
    // under UGVC_POISON the harness itself is checked: a kernel that reads LDS it never wrote must see the pattern on every CU
    const char* pm = getenv("UGVC_POISON");
    const char* pl = getenv("UGVC_POISON_LDS");
    if (!rc && pm && atoi(pm) > 0 && !(pl && pl[0] == 'r')) {
        const uint32_t want = pl ? (uint32_t)strtoul(pl, nullptr, 16) : atoi(pm) == 2 ? 0xFFFFFFFFu : 0xA5A5A5A5u;
        const int n_wg = ctx->n_cus * 4;
        std::vector<uint32_t> w((size_t)n_wg * 8);
        if (lds_probe(ctx, w.data(), n_wg)) return -1;
        for (size_t q = 0; q < w.size(); ++q)
            if (w[q] != want) return fail("selftest: LDS poison did not reach workgroup " + std::to_string(q / 8) + " (the debug harness is not covering every CU)");
    }
    return rc;
}

int ugvc_ref_upload(ugvc_ctx* ctx, const uint8_t* codes, int64_t total_len, const int64_t* contig_off,
                    int n_contigs) {
    if (!ctx || !codes || !contig_off) return fail("NULL argument");
    if (n_contigs < 1 || n_contigs > 65535) return fail("n_contigs must be in 1..65535 (contig column is u16)");
    if (contig_off[0] != 0 || contig_off[n_contigs] != total_len) return fail("contig_off must span [0, total_len]");
    for (int c = 0; c < n_contigs; ++c)
        if (contig_off[c + 1] < contig_off[c]) return fail("contig_off must be non-decreasing");
    UGVC_HIP(hipSetDevice(ctx->device));
    // 64 zero bytes of padding: the v2 kernel reads 16-byte aligned 64-byte windows
    // (and the v4 kernel addresses windows from 32 bytes before a contig's first base: 64 in front too)
    if (ensure(ctx->ref, (size_t)total_len + 64 + kRefFrontPad)) return -1;
    UGVC_HIP(hipMemsetAsync(ctx->ref.p, 0, kRefFrontPad, ctx->stream));
    UGVC_HIP(hipMemsetAsync(static_cast<uint8_t*>(ctx->ref.p) + kRefFrontPad + total_len, 0, 64, ctx->stream));
    if (total_len) UGVC_HIP(copy_in(ctx, static_cast<uint8_t*>(ctx->ref.p) + kRefFrontPad, codes, (size_t)total_len));
    if (upload(ctx, ctx->contig_off, contig_off, sizeof(int64_t) * (n_contigs + 1))) return -1;
    UGVC_HIP(hipStreamSynchronize(ctx->stream));
    ctx->n_contigs = n_contigs;
    ctx->ref_len = total_len;
    return 0;
}

static int check_track(ugvc_ctx* ctx, const int32_t* starts, const int32_t* ends, const int32_t* ptr, int64_t n) {
    if (!ctx) return fail("ctx is NULL");
    if (ctx->n_contigs == 0) return fail("upload the reference before interval tables");
    if (n < 0 || n > std::numeric_limits<int32_t>::max()) return fail("interval count out of range");
    if (n > 0 && (!starts || !ends)) return fail("NULL interval arrays");
    if (!ptr) return fail("NULL contig_ptr");
    if (ptr[0] != 0 || ptr[ctx->n_contigs] != n) return fail("contig_ptr must span [0, n]");
    // the membership test is a pair of binary searches (over starts, over ends): both columns must be
    // non-decreasing inside every contig (sort the BED and merge nested intervals on the host)
    for (int c = 0; c < ctx->n_contigs; ++c) {
        if (ptr[c + 1] < ptr[c]) return fail("contig_ptr must be non-decreasing");
        for (int64_t i = ptr[c]; i + 1 < ptr[c + 1]; ++i)
            if (starts[i] > starts[i + 1] || ends[i] > ends[i + 1])
                return fail("interval table not sorted (starts and ends must be non-decreasing per contig; "
                            "merge nested intervals) at row " + std::to_string(i + 1));
    }
    return 0;
}

int ugvc_runs_upload(ugvc_ctx* ctx, const int32_t* starts, const int32_t* ends, const int32_t* contig_ptr,
                     int64_t n, int min_len, int max_dist, int mark_hpol) {
    if (check_track(ctx, starts, ends, contig_ptr, n)) return -1;
    UGVC_HIP(hipSetDevice(ctx->device));
    // drop runs shorter than min_len (parse_runs_file(runfile, min_hmer_run_length))
    std::vector<int32_t> s, e, p(ctx->n_contigs + 1, 0);
    s.reserve(n);
    e.reserve(n);
    for (int c = 0; c < ctx->n_contigs; ++c) {
        for (int64_t i = contig_ptr[c]; i < contig_ptr[c + 1]; ++i)
            if (ends[i] - starts[i] >= min_len) {
                s.push_back(starts[i]);
                e.push_back(ends[i]);
            }
        p[c + 1] = (int32_t)s.size();
    }
    if (upload(ctx, ctx->runs_s, s.data(), s.size() * 4)) return -1;
    if (upload(ctx, ctx->runs_e, e.data(), e.size() * 4)) return -1;
    if (upload(ctx, ctx->runs_p, p.data(), p.size() * 4)) return -1;
    if (upload_coarse(ctx, ctx->runs_c, s.data(), (int64_t)s.size())) return -1;
    UGVC_HIP(hipStreamSynchronize(ctx->stream));
    ctx->runs_n = (int64_t)s.size();
    ctx->runs_fast = 1;           // the v3 kernel derives #ends < pos from #starts < pos: needs disjoint sorted runs
    for (int c = 0; c < ctx->n_contigs && ctx->runs_fast; ++c)
        for (int64_t i = p[c]; i < p[c + 1]; ++i)
            if (s[i] >= e[i] || (i + 1 < p[c + 1] && e[i] > s[i + 1])) { ctx->runs_fast = 0; break; }
    ctx->has_runs = 1;
    ctx->hpol_dist = max_dist;
    ctx->mark_hpol = mark_hpol;
    return 0;
}

int ugvc_track_upload(ugvc_ctx* ctx, int track_id, const int32_t* starts, const int32_t* ends,
                      const int32_t* contig_ptr, int64_t n) {
    if (track_id < 0 || track_id >= UGVC_MAX_TRACKS) return fail("track_id out of range");
    if (check_track(ctx, starts, ends, contig_ptr, n)) return -1;
    UGVC_HIP(hipSetDevice(ctx->device));
    if (upload(ctx, ctx->trk_s[track_id], starts, (size_t)n * 4)) return -1;
    if (upload(ctx, ctx->trk_e[track_id], ends, (size_t)n * 4)) return -1;
    if (upload(ctx, ctx->trk_p[track_id], contig_ptr, (size_t)(ctx->n_contigs + 1) * 4)) return -1;
    if (upload_coarse(ctx, ctx->trk_c[track_id], starts, n)) return -1;
    UGVC_HIP(hipStreamSynchronize(ctx->stream));
    ctx->trk_n[track_id] = n;
    ctx->trk_fast[track_id] = 1;
    ctx->track_set[track_id] = 1;
    if (track_id + 1 > ctx->n_tracks) ctx->n_tracks = track_id + 1;
    return 0;
}

int ugvc_set_n_tracks(ugvc_ctx* ctx, int n_tracks) {
    if (!ctx) return fail("ctx is NULL");
    if (n_tracks < 0 || n_tracks > UGVC_MAX_TRACKS) return fail("n_tracks out of range");
    ctx->n_tracks = n_tracks;
    return 0;
}

int ugvc_blacklist_upload(ugvc_ctx* ctx, const uint64_t* keys, int64_t n) {
    if (!ctx) return fail("ctx is NULL");
    if (n < 0 || (n > 0 && !keys)) return fail("bad blacklist arguments");
    for (int64_t i = 1; i < n; ++i)
        if (keys[i] <= keys[i - 1]) return fail("blacklist keys must be sorted and unique");
    UGVC_HIP(hipSetDevice(ctx->device));
    if (upload(ctx, ctx->bl, keys, (size_t)n * 8)) return -1;
    if (upload_coarse(ctx, ctx->bl_c, keys, n)) return -1;
    UGVC_HIP(hipStreamSynchronize(ctx->stream));
    ctx->n_bl = n;
    return 0;
}

int ugvc_set_flow_order(ugvc_ctx* ctx, const char* flow4) {
    if (!ctx || !flow4 || strlen(flow4) != 4) return fail("flow order must be 4 characters");
    int seen = 0;
    for (int k = 0; k < 4; ++k) {
        int code = 0;
        switch (flow4[k]) {
            case 'A': case 'a': code = 1; break;
            case 'C': case 'c': code = 2; break;
            case 'G': case 'g': code = 3; break;
            case 'T': case 't': code = 4; break;
            default: return fail("flow order must be a permutation of ACGT");
        }
        seen |= 1 << code;
        ctx->flow[k] = (uint8_t)code;
    }
    if (seen != 0x1e) return fail("flow order must be a permutation of ACGT");
    return build_css_lut(ctx);
}

int ugvc_model_upload(ugvc_ctx* ctx, int group, int kind, const int32_t* feature, const float* threshold,
                      const int32_t* left, const int32_t* right, int32_t n_nodes, const int32_t* tree_root,
                      int32_t n_trees, const double* leaf_value, int32_t n_leaves, int32_t n_features,
                      float base_score, int32_t max_depth) {
    if (!ctx) return fail("ctx is NULL");
    if (group < 0 || group >= UGVC_N_GROUPS) return fail("group out of range");
    if (kind != UGVC_MODEL_RF && kind != UGVC_MODEL_GBT) return fail("unknown model kind");
    if (n_nodes <= 0 || n_trees <= 0 || n_leaves <= 0) return fail("empty model");
    if (n_features > kMaxFeatures) return fail("model has more features than the engine computes");
    UGVC_HIP(hipSetDevice(ctx->device));
    std::vector<Node> nodes(n_nodes);
    for (int i = 0; i < n_nodes; ++i) {
        if (feature[i] < 0) {
            if (left[i] < 0 || left[i] >= n_leaves) return fail("leaf payload index out of range");
            // a leaf loops on itself whichever way the compare goes (a NaN or +inf feature 0 must not leave it);
            // its payload row travels in the bits of `thr`
            nodes[i] = Node{__builtin_bit_cast(float, (int32_t)left[i]), 0, i, i};
        } else {
            if (feature[i] >= n_features) return fail("node feature index out of range");
            if (left[i] < 0 || left[i] >= n_nodes || right[i] < 0 || right[i] >= n_nodes)
                return fail("child index out of range");
            nodes[i] = Node{threshold[i], feature[i], left[i], right[i]};
        }
    }
    // depth check (also guards against cycles)
    int depth = 0;
    {
        std::vector<std::pair<int, int>> stack;
        for (int t = 0; t < n_trees; ++t) {
            if (tree_root[t] < 0 || tree_root[t] >= n_nodes) return fail("tree root out of range");
            stack.push_back({tree_root[t], 0});
            int64_t visited = 0;
            while (!stack.empty()) {
                auto [i, d] = stack.back();
                stack.pop_back();
                if (++visited > (int64_t)n_nodes) return fail("tree structure has a cycle");
                if (feature[i] < 0) {
                    depth = std::max(depth, d);
                } else {
                    stack.push_back({left[i], d + 1});
                    stack.push_back({right[i], d + 1});
                }
            }
        }
    }
    if (max_depth > 0 && max_depth != depth) return fail("max_depth does not match the node table");
    auto& m = ctx->model[group];
    if (upload(ctx, m.nodes, nodes.data(), nodes.size() * sizeof(Node))) return -1;
    if (upload(ctx, m.roots, tree_root, (size_t)n_trees * 4)) return -1;
    if (upload(ctx, m.leaves, leaf_value, (size_t)n_leaves * 16)) return -1;
    UGVC_HIP(hipStreamSynchronize(ctx->stream));
    m.n_trees = n_trees;
    m.depth = depth;
    m.kind = kind;
    m.n_features = n_features;
    m.base = base_score;
    m.set = 1;
    m.has_dense = 0;
    return pack_model_group(ctx, group, feature, threshold, left, right, n_nodes, tree_root, n_trees, leaf_value,
                            n_leaves, n_features, kind, base_score, depth);
}

int ugvc_model_clear(ugvc_ctx* ctx, int group) {
    if (!ctx) return fail("ctx is NULL");
    if (group < 0 || group >= UGVC_N_GROUPS) return fail("group out of range");
    UGVC_HIP(hipSetDevice(ctx->device));
    UGVC_HIP(hipStreamSynchronize(ctx->stream));
    ctx->model[group].set = 0;
    return clear_model_group(ctx, group);
}

static int check_variants(const ugvc_variants* v) {
    if (!v) return fail("variants is NULL");
    if (v->n < 0) return fail("negative variant count");
    if (v->n == 0) return 0;
    if (!v->contig || !v->pos || !v->ref_len || !v->alt_len || !v->ref_off || !v->alt_off || !v->alleles ||
        !v->qual || !v->sor || !v->dp || !v->ad_ref || !v->ad_alt || !v->gq)
        return fail("NULL variant column");
    return 0;
}

int ugvc_variants_upload(ugvc_ctx* ctx, const ugvc_variants* v) {
    if (!ctx) return fail("ctx is NULL");
    if (check_variants(v)) return -1;
    if (ctx->n_contigs == 0) return fail("upload the reference before variants");
    UGVC_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)v->n;
    // host-side validation the kernels rely on (sortedness, contig range, allele bounds): host_rows.cpp, vectorised
    int64_t n_indel = 0, bad = -1;                             // (n_indel sizes the indel tiles' table slices: model_pack.hip)
    if (const int what = validate_rows(v, 0, (int64_t)n, ctx->n_contigs, &n_indel, &bad))
        return fail(std::string(row_error_text(what)) + std::to_string(bad));
    // from here on the resident columns change: on any error below the context is EMPTY, not the previous callset's row
    // count beside re-allocated columns
    UGVC_HIP(hipStreamSynchronize(ctx->stream));
    ctx->n = 0; ctx->n_indel = 0; ctx->scored = 0; ctx->density_n = 0;
    if (upload(ctx, ctx->v_contig, v->contig, n * 2)) return -1;
    if (upload(ctx, ctx->v_pos, v->pos, n * 4)) return -1;
    if (upload(ctx, ctx->v_rl, v->ref_len, n * 2)) return -1;
    if (upload(ctx, ctx->v_al, v->alt_len, n * 2)) return -1;
    if (upload(ctx, ctx->v_ro, v->ref_off, n * 4)) return -1;
    if (upload(ctx, ctx->v_ao, v->alt_off, n * 4)) return -1;
    // 16 zero bytes behind the allele pool: allele tails are fetched with fixed-width loads
    if (ensure(ctx->v_alleles, (size_t)v->alleles_len + 16)) return -1;
    UGVC_HIP(hipMemsetAsync(static_cast<uint8_t*>(ctx->v_alleles.p) + v->alleles_len, 0, 16, ctx->stream));
    if (upload(ctx, ctx->v_alleles, v->alleles, (size_t)v->alleles_len)) return -1;
    if (upload(ctx, ctx->v_qual, v->qual, n * 4)) return -1;
    if (upload(ctx, ctx->v_sor, v->sor, n * 4)) return -1;
    if (upload(ctx, ctx->v_dp, v->dp, n * 4)) return -1;
    if (upload(ctx, ctx->v_adr, v->ad_ref, n * 4)) return -1;
    if (upload(ctx, ctx->v_ada, v->ad_alt, n * 4)) return -1;
    if (upload(ctx, ctx->v_gq, v->gq, n)) return -1;
    if (ensure(ctx->r_score, n * 4) || ensure(ctx->r_filter, n) || ensure(ctx->r_flags, n)) return -1;
    UGVC_HIP(hipStreamSynchronize(ctx->stream));
    ctx->n = v->n;
    ctx->n_indel = n_indel;
    ctx->scored = 0;
    return 0;
}

int ugvc_filter_resident(ugvc_ctx* ctx) {
    if (!ctx) return fail("ctx is NULL");
    UGVC_HIP(hipSetDevice(ctx->device));
    FilterArgs a;
    if (build_args(ctx, a, false)) return -1;
    return launch_score(ctx, a);
}

int ugvc_results_download(ugvc_ctx* ctx, const ugvc_results* out) {
    if (!ctx || !out) return fail("NULL argument");
    UGVC_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)ctx->n;
    if (n) {
        if (out->tree_score) UGVC_HIP(copy_out(ctx, out->tree_score, ctx->r_score.p, n * 4));
        if (out->filter) UGVC_HIP(copy_out(ctx, out->filter, ctx->r_filter.p, n));
        if (out->flags) UGVC_HIP(copy_out(ctx, out->flags, ctx->r_flags.p, n));
    }
    UGVC_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int ugvc_filter_variants(ugvc_ctx* ctx, const ugvc_variants* v, const ugvc_results* out) {
    // large callsets: a chunk pipeline - host validation + staging || H2D || pass || D2H (pipeline.hip); UGVC_PIPE_CHUNKS
    // sets the number of chunks (1 = the plain upload / pass / download sequence below)
    if (ctx && v && out && v->n >= 262144 && check_variants(v) == 0 && ctx->n_contigs > 0) {
        const char* e = getenv("UGVC_PIPE_CHUNKS");          // (read per call: a sweep changes it inside one process)
        const int chunks = std::min(e ? atoi(e) : 8, 64);
        if (chunks > 1) {
            UGVC_HIP(hipSetDevice(ctx->device));
            return filter_variants_pipelined(ctx, v, out, chunks);
        }
    }
    if (ugvc_variants_upload(ctx, v)) return -1;
    if (ugvc_filter_resident(ctx)) return -1;
    return ugvc_results_download(ctx, out);
}

int ugvc_reserve(ugvc_ctx* ctx, int64_t n_variants, int64_t alleles_len) {
    // Everything ugvc_filter_variants allocates once per size - resident columns, pinned staging slots, streams, events, the
    // worker pool - and the first use of the kernels' code object, without touching a row of the CALLER's: a resident callset
    // survives only if no resident column has to grow (else the context is left empty - filter_variants_pipelined).  A tool calls it from a helper
    // thread as soon as it knows the callset's size, beside its other set-up work (the first pass over 5 M rows spent 44 of
    // its 49 ms there).  Safe beside uploads of the reference / tables / model on another thread; not beside a pass.
    if (!ctx) return fail("ctx is NULL");
    if (n_variants < 0 || alleles_len < 0) return fail("bad reserve sizes");
    UGVC_HIP(hipSetDevice(ctx->device));
    if (v5_warm(ctx)) return -1;
    if (n_variants < 262144) return 0;                           // (small callsets take the plain upload path: nothing to prepare)
    const char* e = getenv("UGVC_PIPE_CHUNKS");
    const int chunks = std::min(e ? atoi(e) : 8, 64);
    if (chunks <= 1) return 0;
    ugvc_variants v;
    memset(&v, 0, sizeof v);
    v.n = n_variants;
    v.alleles_len = alleles_len;
    return filter_variants_pipelined(ctx, &v, nullptr, chunks, true);
}

int ugvc_resident_count(ugvc_ctx* ctx, int64_t* n_variants, int* scored) {
    if (!ctx) return fail("ctx is NULL");
    if (n_variants) *n_variants = ctx->n;
    if (scored) *scored = ctx->scored;
    return 0;
}

int ugvc_timed_filter(ugvc_ctx* ctx, int iters, float* ms_total) {
    if (!ctx || !ms_total || iters < 1) return fail("bad arguments");
    UGVC_HIP(hipSetDevice(ctx->device));
    FilterArgs a;
    if (build_args(ctx, a, false)) return -1;
    UGVC_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    for (int it = 0; it < iters; ++it)
        if (launch_score(ctx, a)) return -1;
    UGVC_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    UGVC_HIP(hipEventSynchronize(ctx->ev1));
    UGVC_HIP(hipEventElapsedTime(ms_total, ctx->ev0, ctx->ev1));
    return 0;
}

int ugvc_device_sync(ugvc_ctx* ctx) {
    if (!ctx) return fail("ctx is NULL");
    UGVC_HIP(hipSetDevice(ctx->device));
    UGVC_HIP(hipDeviceSynchronize());
    return 0;
}

int ugvc_allgather_resident(ugvc_ctx* ctx, int64_t shard_cap);
int ugvc_gather_fence(ugvc_ctx* ctx);
int ugvc_gather_target(ugvc_ctx* ctx, int64_t shard_cap, float** score, uint8_t** filter, uint8_t** flags);
int ugvc_gather_launch(ugvc_ctx* ctx, int64_t shard_cap);

int ugvc_pass_clock(ugvc_ctx* ctx, int passes, double* shader_ghz, double* wave_ms) {
    // The shader clock the scoring pass actually runs at: `passes` resident passes back to back (the clock state settles over
    // tens of dispatches), the last of them probed - workgroup 0's first wave reads the constant 100 MHz counter (s_memrealtime)
    // and the shader-clock counter (s_memtime) at its entry and at its end; GHz = shader ticks / (100 MHz ticks x 10 ns).
    if (!ctx || !shader_ghz || passes < 1) return fail("bad arguments");
    UGVC_HIP(hipSetDevice(ctx->device));
    FilterArgs a;
    if (build_args(ctx, a, false)) return -1;
    if (!(!(ctx->kernel_variant & 256) && v2_available(ctx) && !(ctx->kernel_variant & 65536) && v5_available(ctx)))
        return fail("ugvc_pass_clock: the clock words are written by the v5 pass, which this configuration does not take");
    for (int it = 0; it + 1 < passes; ++it)
        if (launch_score(ctx, a)) return -1;
    ctx->clk_probe = 1;
    ctx->clk_rt = ctx->clk_sh = 0;
    const int rc = launch_score(ctx, a);
    ctx->clk_probe = 0;
    if (rc) return -1;
    UGVC_HIP(hipStreamSynchronize(ctx->stream));
    if (!ctx->clk_rt || !ctx->clk_sh) return fail("ugvc_pass_clock: no clock words came back (no rows resident, or the pass did not run the v5 kernel)");
    *shader_ghz = (double)ctx->clk_sh / ((double)ctx->clk_rt * 10.0);
    if (wave_ms) *wave_ms = (double)ctx->clk_rt * 1e-5;
    return 0;
}

int ugvc_set_step_events(ugvc_ctx* ctx, int on) {
    if (!ctx) return fail("ctx is NULL");
    ctx->step_events = on ? 1 : 0;
    return 0;
}

int ugvc_timed_steps(ugvc_ctx* ctx, int iters, int64_t shard_cap, int gather, float* ms_total, float* ms_kernel) {
    if (!ctx || !ms_total || !ms_kernel || iters < 1) return fail("bad arguments");
    UGVC_HIP(hipSetDevice(ctx->device));
    FilterArgs a;
    if (build_args(ctx, a, false)) return -1;
    // Per-step event pairs are a measurement aid with a price: an event record is a marker packet between two launches (the
    // stream drains to it) - ~9 us of every step of a 5 M pass, ~10 % of a 625 k-variant shard's.  ugvc_set_step_events(ctx, 0):
    // ONE pair around the whole run, the passes back to back as a production stream issues them; *ms_kernel is then the time
    // between that pair (without a collective: exactly the launches) and every step's entry of ugvc_last_step_ms its mean.
    const bool per_step = ctx->step_events != 0;
    std::vector<hipEvent_t> ev(per_step ? 2 * (size_t)iters : 0);
    for (auto& e : ev) UGVC_HIP(hipEventCreate(&e));
    int rc = 0;
    UGVC_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    for (int it = 0; it < iters && !rc; ++it) {
        // with a collective the pass writes straight into this step's gather buffer (no copies)
        if (gather) rc = ugvc_gather_target(ctx, shard_cap, &a.score, &a.filter, &a.flags);
        if (rc) break;
        if (per_step) UGVC_HIP(hipEventRecord(ev[2 * it], ctx->stream));
        rc = launch_score(ctx, a);
        if (per_step) UGVC_HIP(hipEventRecord(ev[2 * it + 1], ctx->stream));
        if (!rc && gather) rc = ugvc_gather_launch(ctx, shard_cap);
    }
    if (!rc && gather) rc = ugvc_gather_fence(ctx);      // the timed region ends when the last gather has landed
    UGVC_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    UGVC_HIP(hipEventSynchronize(ctx->ev1));
    if (!rc) {
        UGVC_HIP(hipEventElapsedTime(ms_total, ctx->ev0, ctx->ev1));
        float sum = 0.f;
        ctx->step_ms.assign((size_t)iters, *ms_total / (float)iters);
        if (per_step) {
            for (int it = 0; it < iters; ++it) {
                float ms = 0.f;
                UGVC_HIP(hipEventElapsedTime(&ms, ev[2 * it], ev[2 * it + 1]));
                ctx->step_ms[(size_t)it] = ms;
                sum += ms;
            }
        } else sum = *ms_total;
        *ms_kernel = sum;
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    return rc;
}

int ugvc_last_step_ms(ugvc_ctx* ctx, float* out, int cap) {
    if (!ctx || (cap > 0 && !out)) return fail("NULL argument");
    const int n = (int)std::min<size_t>(ctx->step_ms.size(), (size_t)std::max(cap, 0));
    for (int k = 0; k < n; ++k) out[k] = ctx->step_ms[(size_t)k];
    return n;
}

int ugvc_n_features(ugvc_ctx* ctx) { return ctx ? UGVC_N_BASE_FEATURES + ctx->n_tracks : -1; }

int ugvc_debug_phase_clocks(ugvc_ctx* ctx, uint64_t out[8], int reset) {
    if (!ctx || !out) return fail("NULL argument");
    return v2_phase_clocks(ctx, out, reset);
}

int ugvc_set_kernel_variant(ugvc_ctx* ctx, int variant) {
    if (!ctx) return fail("ctx is NULL");
    ctx->kernel_variant = variant;
    return 0;
}

int ugvc_feature_matrix(ugvc_ctx* ctx, float* x_host, uint8_t* group_host) {
    if (!ctx || !x_host) return fail("NULL argument");
    UGVC_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)ctx->n;
    const size_t F = (size_t)(UGVC_N_BASE_FEATURES + ctx->n_tracks);
    if (ensure(ctx->x_mat, n * F * 4) || ensure(ctx->x_group, n)) return -1;
    FilterArgs a;
    if (build_args(ctx, a, true)) return -1;
    // the fused kernel's featurize waves (kernels_v5.hip, WX) when the side tables allow, else the universal kernel
    if (fm5_available(ctx) ? launch_feature_matrix_v5(ctx, a) : launch_filter(ctx, a, false, true)) return -1;
    if (n) {
        UGVC_HIP(copy_out(ctx, x_host, ctx->x_mat.p, n * F * 4));
        if (group_host) UGVC_HIP(copy_out(ctx, group_host, ctx->x_group.p, n));
    }
    UGVC_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}


// `iters` back-to-back builds of the resident N x F feature matrix (no download): bench.py's C5 "feature-build GB/s"
int ugvc_timed_feature_matrix(ugvc_ctx* ctx, int iters, float* ms_total) {
    if (!ctx || !ms_total || iters < 1) return fail("bad arguments");
    UGVC_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)ctx->n;
    const size_t F = (size_t)(UGVC_N_BASE_FEATURES + ctx->n_tracks);
    if (ensure(ctx->x_mat, n * F * 4) || ensure(ctx->x_group, n)) return -1;
    FilterArgs a;
    if (build_args(ctx, a, true)) return -1;
    UGVC_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    for (int it = 0; it < iters; ++it)
        if (fm5_available(ctx) ? launch_feature_matrix_v5(ctx, a) : launch_filter(ctx, a, false, true)) return -1;
    UGVC_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    UGVC_HIP(hipEventSynchronize(ctx->ev1));
    UGVC_HIP(hipEventElapsedTime(ms_total, ctx->ev0, ctx->ev1));
    return 0;
}

}  // extern "C"
