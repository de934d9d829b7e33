// Fused featurize -> lookup -> score -> FILTER kernel for gfx950 (MI355X).
//
// One thread per variant, 256 variants per workgroup.  Semantics follow the CPU oracle
// (oracle/oracle.py), which restates - by cited call site - the reference's
//   classify_indel / is_hmer_indel / get_motif_around   (ugvc/pipelines/run_no_gt_report.py:92-94)
//   gc_content / interval columns                       (ugvc/reports/report_data_loader.py:67-94)
//   close_to_hmer_run (--hpol_filter_length_dist)       (docs/filter_variants_pipeline.md:30-33)
//   cycle-skip status (--flow_order)                    (docs/filter_variants_pipeline.md:43-44)
//   blacklist -> COHORT_FP, model -> TREE_SCORE, PASS/LOW_SCORE (docs/howto-callset-filter.md:61-65)
//
// Layout: every variant column is a contiguous array (coalesced 1-16 B/lane loads); the
// reference genome is 1 byte/base; features of the block are staged in LDS as xs[f][lane]
// (bank = lane % 32 for every f, so the per-lane dynamic feature index of a tree walk is
// conflict-free); tree nodes are 16-byte records walked with self-looping leaves (no leaf
// branch, trees advance in lock-step across the wave, 4 trees in flight per lane for ILP).
#include "ugvc_device.hpp"

namespace ugvc {

__device__ __forceinline__ int lower_bound_i32(const int32_t* __restrict__ a, int lo, int hi, int key) {
    // number of elements < key in a[lo:hi)  (numpy searchsorted side='left'), returned as an
    // index relative to lo
    int base = lo, len = hi - lo;
    while (len > 0) {
        int half = len >> 1;
        int mid = base + half;
        bool lt = a[mid] < key;
        base = lt ? mid + 1 : base;
        len = lt ? len - half - 1 : half;
    }
    return base - lo;
}

__device__ __forceinline__ bool contains_u64(const uint64_t* __restrict__ a, int64_t n, uint64_t key) {
    int64_t base = 0, len = n;
    while (len > 0) {
        int64_t half = len >> 1;
        int64_t mid = base + half;
        bool lt = a[mid] < key;
        base = lt ? mid + 1 : base;
        len = lt ? len - half - 1 : half;
    }
    return base < n && a[base] == key;
}

__device__ __forceinline__ bool inside_track(const TrackView& t, int c, int pos) {
    int lo = t.ptr[c], hi = t.ptr[c + 1];
    int s = lower_bound_i32(t.starts, lo, hi, pos) - 1;
    int e = lower_bound_i32(t.ends, lo, hi, pos);
    return s == e;
}

struct RefWin {
    const uint8_t* __restrict__ codes;
    int64_t lo, hi;   // contig bounds in the concatenated genome
    __device__ __forceinline__ int at(int64_t i) const { return (i >= lo && i < hi) ? codes[i] : 0; }
};

// flow-key walk of two equal-length sequences in lock-step (oracle.cycle_skip_status)
template <class SeqR, class SeqA>
__device__ __forceinline__ int cycle_skip(int L, const uint8_t flow[4], SeqR seq_r, SeqA seq_a) {
    int pr = 0, pa = 0, lr = 0, la = 0;
    bool poss = false;
    for (int s = 0; pr < L || pa < L; ++s) {
        int b = flow[s & 3];
        bool ar = pr < L, aa = pa < L;
        int hr = 0, ha = 0;
        if (ar) {
            while (pr + hr < L && seq_r(pr + hr) == b) ++hr;
            pr += hr;
            ++lr;
        }
        if (aa) {
            while (pa + ha < L && seq_a(pa + ha) == b) ++ha;
            pa += ha;
            ++la;
        }
        if (ar && aa && hr != ha && (hr == 0 || ha == 0)) poss = true;
    }
    if (lr != la) return 2;
    return poss ? 1 : 0;
}

template <int KIND>
__device__ __forceinline__ void walk_forest(const ForestView& f, const float* __restrict__ xs_lane,
                                            float& score, uint8_t& filt) {
    // xs_lane points at xs[0][lane]; feature f of this lane is xs_lane[f * kBlock]
    const Node* __restrict__ nodes = f.nodes;
    const int* __restrict__ roots = f.roots;
    const double2* __restrict__ leaves = f.leaves;
    double a0 = 0.0, a1 = 0.0;
    float margin = f.base;
    const int T = f.n_trees, D = f.depth;
    int t = 0;
    for (; t + 4 <= T; t += 4) {
        int i0 = roots[t], i1 = roots[t + 1], i2 = roots[t + 2], i3 = roots[t + 3];
        for (int d = 0; d < D; ++d) {
            Node n0 = nodes[i0], n1 = nodes[i1], n2 = nodes[i2], n3 = nodes[i3];
            float x0 = xs_lane[n0.feat * kBlock], x1 = xs_lane[n1.feat * kBlock];
            float x2 = xs_lane[n2.feat * kBlock], x3 = xs_lane[n3.feat * kBlock];
            if (KIND == UGVC_MODEL_RF) {
                i0 = x0 <= n0.thr ? n0.left : n0.right;
                i1 = x1 <= n1.thr ? n1.left : n1.right;
                i2 = x2 <= n2.thr ? n2.left : n2.right;
                i3 = x3 <= n3.thr ? n3.left : n3.right;
            } else {
                i0 = x0 < n0.thr ? n0.left : n0.right;
                i1 = x1 < n1.thr ? n1.left : n1.right;
                i2 = x2 < n2.thr ? n2.left : n2.right;
                i3 = x3 < n3.thr ? n3.left : n3.right;
            }
        }
        double2 v0 = leaves[__float_as_int(nodes[i0].thr)], v1 = leaves[__float_as_int(nodes[i1].thr)];
        double2 v2 = leaves[__float_as_int(nodes[i2].thr)], v3 = leaves[__float_as_int(nodes[i3].thr)];
        if (KIND == UGVC_MODEL_RF) {   // strict tree order, f64 (== sklearn predict_proba)
            a0 += v0.x; a1 += v0.y; a0 += v1.x; a1 += v1.y;
            a0 += v2.x; a1 += v2.y; a0 += v3.x; a1 += v3.y;
        } else {
            margin += (float)v0.x; margin += (float)v1.x; margin += (float)v2.x; margin += (float)v3.x;
        }
    }
    for (; t < T; ++t) {
        int i0 = roots[t];
        for (int d = 0; d < D; ++d) {
            Node n0 = nodes[i0];
            float x0 = xs_lane[n0.feat * kBlock];
            if (KIND == UGVC_MODEL_RF) i0 = x0 <= n0.thr ? n0.left : n0.right;
            else i0 = x0 < n0.thr ? n0.left : n0.right;
        }
        double2 v0 = leaves[__float_as_int(nodes[i0].thr)];
        if (KIND == UGVC_MODEL_RF) { a0 += v0.x; a1 += v0.y; }
        else margin += (float)v0.x;
    }
    if (KIND == UGVC_MODEL_RF) {
        double p0 = a0 / (double)T, p1 = a1 / (double)T;
        score = (float)p1;
        filt = p1 > p0 ? UGVC_FILTER_PASS : UGVC_FILTER_LOW_SCORE;
    } else {
        score = 1.0f / (1.0f + expf(-margin));
        filt = margin > 0.0f ? UGVC_FILTER_PASS : UGVC_FILTER_LOW_SCORE;
    }
}

template <bool SCORE, bool WRITE_X>
__global__ __launch_bounds__(kBlock) void filter_kernel(const FilterArgs a) {
    __shared__ float xs[kMaxFeatures * kBlock];
    const int tid = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * kBlock + tid;
    const bool live = i < a.n;
    const int F = UGVC_N_BASE_FEATURES + a.n_tracks;
    int group = 0;
    uint8_t flags = 0;

    if (live) {
        const int c = a.contig[i];
        const int pos = a.pos[i];
        const int rl = a.ref_len[i], al = a.alt_len[i];
        const uint32_t ro = a.ref_off[i], ao = a.alt_off[i];
        const uint8_t* __restrict__ pool = a.alleles;
        RefWin rw{a.ref, a.contig_off[c], a.contig_off[c + 1]};
        const int64_t g0 = rw.lo + pos - 1;

        // ---- classify_indel
        const bool indel = rl != al;
        const int classify = !indel ? 0 : (rl < al ? 1 : 2);
        const int indel_length = rl < al ? al - rl : rl - al;

        // ---- is_hmer_indel
        int hmer_len = 0, hmer_nuc = 0;
        if (indel && !(a.ablate & 8)) {
            const uint8_t* alle = pool + (classify == 1 ? ao : ro);
            const int ln = classify == 1 ? al : rl;
            const int b = alle[1];
            bool mono = true;
            for (int k = 2; k < ln; ++k) mono &= alle[k] == b;
            const int64_t start = classify == 1 ? g0 + 1 : g0 + rl;
            if (mono && start >= rw.lo && start < rw.hi && rw.codes[start] == b) {
                int64_t j = start + 1;
                while (j < rw.hi && rw.codes[j] == b) ++j;
                hmer_len = (int)(j - start) + (classify == 1 ? 0 : rl - 1);
                hmer_nuc = b;
            }
        }
        const bool is_h = indel && hmer_len > 0;
        group = !indel ? 0 : (is_h ? 1 : 2);

        // ---- get_motif_around (size 5)
        const int64_t lstart = indel ? g0 - (kMotif - 1) : g0 - kMotif;
        const int64_t rstart = !indel ? g0 + 1 : (is_h ? g0 + 1 + hmer_len : g0 + rl);
        int lmb[kMotif], rmb[kMotif];
        int lm = 0, rm = 0;
#pragma unroll
        for (int k = 0; k < kMotif; ++k) {
            lmb[k] = (a.ablate & 8) ? 1 : rw.at(lstart + k);
            rmb[k] = (a.ablate & 8) ? 2 : rw.at(rstart + k);
            lm = lm * 5 + lmb[k];
            rm = rm * 5 + rmb[k];
        }

        // ---- gc_content (window 10 starting at pos - 5, 0-based slice of the 1-based pos)
        int gc_cnt = 0, gc_len = 0;
#pragma unroll
        for (int k = 0; k < kGcWindow; ++k) {
            const int64_t w = g0 + 1 - kGcWindow / 2 + k;
            const bool inb = w >= rw.lo && w < rw.hi && !(a.ablate & 8);
            const int b = inb ? rw.codes[w] : 0;
            gc_len += inb;
            gc_cnt += inb && b != 1 && b != 4;
        }
        const float gc = gc_len > 0 ? (float)((double)gc_cnt / (double)gc_len) : 0.0f;

        // ---- cycle skip (substitutions only)
        int css = 3;
        if (!indel && !(a.ablate & 4)) {
            const int L = rl + 2 * kMotif;
            bool has_n = false;
#pragma unroll
            for (int k = 0; k < kMotif; ++k) has_n |= lmb[k] == 0 || rmb[k] == 0;
            for (int k = 0; k < rl; ++k) has_n |= pool[ro + k] == 0 || pool[ao + k] == 0;
            if (has_n) {
                css = 0;
            } else {
                uint32_t lpk = 0, rpk = 0;   // 3 bits per base
#pragma unroll
                for (int k = 0; k < kMotif; ++k) {
                    lpk |= (uint32_t)lmb[k] << (3 * k);
                    rpk |= (uint32_t)rmb[k] << (3 * k);
                }
                auto seq_r = [&](int k) -> int {
                    if (k < kMotif) return (lpk >> (3 * k)) & 7;
                    if (k < kMotif + rl) return pool[ro + k - kMotif];
                    return (rpk >> (3 * (k - kMotif - rl))) & 7;
                };
                auto seq_a = [&](int k) -> int {
                    if (k < kMotif) return (lpk >> (3 * k)) & 7;
                    if (k < kMotif + rl) return pool[ao + k - kMotif];
                    return (rpk >> (3 * (k - kMotif - rl))) & 7;
                };
                css = cycle_skip(L, a.flow, seq_r, seq_a);
            }
        }

        // ---- hmer-run proximity (close_to_hmer_run) and interval tracks
        bool inside_run = false, close_run = false;
        if (a.has_runs && !(a.ablate & 2)) {
            const int lo = a.runs.ptr[c], hi = a.runs.ptr[c + 1];
            const int nr = hi - lo;
            if (nr > 0) {
                const int32_t* __restrict__ st = a.runs.starts + lo;
                const int32_t* __restrict__ en = a.runs.ends + lo;
                const int s = lower_bound_i32(st, 0, nr, pos) - 1;
                const int e = lower_bound_i32(en, 0, nr, pos);
                const int64_t p = pos, D = a.hpol_dist;
                auto near = [&](int64_t x) { int64_t d = p - x; return (d < 0 ? -d : d) < D; };
                bool cd = near(st[s < 0 ? 0 : s]);
                cd |= near(st[s + 1 > nr - 1 ? nr - 1 : s + 1]);
                cd |= near(en[e - 1 < 0 ? 0 : e - 1]);
                cd |= near(en[e > nr - 1 ? nr - 1 : e]);
                inside_run = s == e;
                close_run = cd && !inside_run;
            }
        }
        if (a.mark_hpol && (inside_run || close_run)) flags |= UGVC_FLAG_HPOL_RUN;
        float trk[UGVC_MAX_TRACKS];
#pragma unroll
        for (int t = 0; t < UGVC_MAX_TRACKS; ++t) {
            trk[t] = 0.f;
            if (t < a.n_tracks && !(a.ablate & 2)) {
                const bool in = inside_track(a.tracks[t], c, pos);
                trk[t] = in ? 1.f : 0.f;
                flags |= in ? (uint8_t)(1u << (UGVC_FLAG_TRACK0_SHIFT + t)) : 0;
            }
        }
        if (a.n_bl > 0 && !(a.ablate & 2) && contains_u64(a.bl, a.n_bl, ((uint64_t)c << 32) | (uint32_t)pos))
            flags |= UGVC_FLAG_COHORT_FP;

        // ---- feature vector (schema.BASE_FEATURES order) into LDS
        const int dp = a.dp[i], adr = a.ad_ref[i], ada = a.ad_alt[i];
        const float vaf = dp > 0 ? __fdiv_rn((float)ada, (float)dp) : 0.0f;
        float* x = xs + tid;
        x[0 * kBlock] = a.qual[i];
        x[1 * kBlock] = a.sor[i];
        x[2 * kBlock] = (float)dp;
        x[3 * kBlock] = (float)adr;
        x[4 * kBlock] = (float)ada;
        x[5 * kBlock] = vaf;
        x[6 * kBlock] = (float)a.gq[i];
        x[7 * kBlock] = (float)classify;
        x[8 * kBlock] = (float)indel_length;
        x[9 * kBlock] = (float)hmer_len;
        x[10 * kBlock] = (float)hmer_nuc;
        x[11 * kBlock] = (float)lm;
        x[12 * kBlock] = (float)rm;
        x[13 * kBlock] = gc;
        x[14 * kBlock] = (float)css;
        x[15 * kBlock] = inside_run ? 1.f : 0.f;
        x[16 * kBlock] = close_run ? 1.f : 0.f;
#pragma unroll
        for (int t = 0; t < UGVC_MAX_TRACKS; ++t)
            if (t < a.n_tracks) x[(UGVC_N_BASE_FEATURES + t) * kBlock] = trk[t];
    }

    if (WRITE_X) {
        __syncthreads();
        const int64_t row0 = (int64_t)blockIdx.x * kBlock;
        const int64_t rows = a.n - row0 < kBlock ? a.n - row0 : kBlock;
        const int64_t total = rows * F;
        float* __restrict__ out = a.X + row0 * F;
        for (int64_t j = tid; j < total; j += kBlock) {
            const int r = (int)(j / F), f = (int)(j - (int64_t)r * F);
            out[j] = xs[f * kBlock + r];
        }
        if (live && a.group) a.group[i] = (uint8_t)group;
    }

    if (SCORE && live) {
        float score = 0.f;
        uint8_t filt = UGVC_FILTER_PASS;
        // each lane only reads its own LDS column, written above by itself: no barrier needed
        const ForestView& f = a.forest[group];
        if (f.n_trees > 0 && !(a.ablate & 1)) {
            if (f.kind == UGVC_MODEL_RF) walk_forest<UGVC_MODEL_RF>(f, xs + tid, score, filt);
            else walk_forest<UGVC_MODEL_GBT>(f, xs + tid, score, filt);
        }
        a.score[i] = score;
        a.filter[i] = filt;
        a.flags[i] = flags;
    }
}

int launch_filter(ugvc_ctx* ctx, const FilterArgs& a, bool score, bool write_x) {
    if (a.n == 0) return 0;
    const unsigned grid = (unsigned)((a.n + kBlock - 1) / kBlock);
    if (score && write_x) UGVC_LAUNCH((filter_kernel<true, true>), dim3(grid), dim3(kBlock), 0, ctx->stream, a);
    else if (score) UGVC_LAUNCH((filter_kernel<true, false>), dim3(grid), dim3(kBlock), 0, ctx->stream, a);
    else UGVC_LAUNCH((filter_kernel<false, true>), dim3(grid), dim3(kBlock), 0, ctx->stream, a);
    UGVC_HIP(hipGetLastError());
    return 0;
}

}  // namespace ugvc
