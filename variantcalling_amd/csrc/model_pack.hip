// Host-side preparation of the v2 fast path: rank-quantised dense forests, per-feature code
// tables (LUT / sorted thresholds), code bit-packing layout, cycle-skip LUT.  See ugvc_v2.hpp.
#include <algorithm>
#include <cmath>
#include <limits>
#include <map>

#include "ugvc_v2.hpp"

namespace ugvc {

struct V2Group {
    bool set = false, ok = false;
    int kind = 0, T = 0, D = 0, n_pairs = 0, n_features = 0;
    float base = 0.f;
    std::string why;
    std::vector<uint32_t> nodes;       // filled by finalize (needs the packing layout)
    std::vector<uint16_t> leaf_idx;
    std::vector<double> pairs;
    std::vector<float> leaf_f32;
    std::vector<uint32_t> hi5, last5, roots5;   // v5: raw-code node tables, one heap over all trees (pack_group5)
    int cap5[kMaxFeatures];
    uint32_t used5 = 0;
    bool ok5 = false;
    DeviceBuf d_hi5, d_last5, d_roots5;
    std::vector<uint32_t> hi4;          // single-sum layout (ugvc_v2.hpp)
    std::vector<uint32_t> last4;        // 2 dwords per entry
    std::vector<double> p1;
    bool fast4 = false;
    double band4 = 0.0;
    std::vector<std::vector<float>> uthr;   // sorted unique thresholds per feature
    // pointer-layout copy kept to rebuild the dense tables when the layout changes
    std::vector<int32_t> feature, left, right, roots;
    std::vector<float> threshold;
    std::vector<double> leaf_value;
    int dword[kMaxFeatures], bit_off[kMaxFeatures], width[kMaxFeatures], plane[kMaxFeatures];
    int jdword[kMaxFeatures], jbit[kMaxFeatures];
    int n_planes = 0;
    bool layout_done = false;               // dword / bit_off already chosen by the joint (group-uniform) layout
    std::vector<uint32_t> plane_desc;
    DeviceBuf d_nodes, d_leaf_idx, d_pairs, d_leaf_f32, d_plane_desc, d_hi4, d_last4, d_p1;
};

struct V2State {
    V2Group g[UGVC_N_GROUPS];
    DeviceBuf desc, desc3, lut, thr, css, brackets, brackets3, counters, prof;
    DeviceBuf snp_idx, indel_idx, counters5, rec5[UGVC_N_GROUPS], eyt, gcr;   // v5
    int c5_set = 0;                          // v5 record counters: two sets, alternating passes
    bool c5_dirty[2] = {true, true};
    int eyt_off[3] = {0, 0, 0}, eyt_bits[3] = {0, 0, 0}, eyt_len = 0;
    int thr0_len = 0, thr0_bits4[4] = {0, 0, 0, 0};
    std::vector<uint2> h_desc3;
    int thr_bits4[4] = {0, 0, 0, 0};        // descent depth per float feature (qual, sor, vaf, gc), max over groups
    bool uniform_layout = false;            // every group uses the same dword per feature; booleans in fixed slots
    int dw3[kMaxFeatures];                  // that dword
    uint32_t boolmask3[UGVC_N_GROUPS] = {0, 0, 0};
    DeviceBuf rec[UGVC_N_GROUPS];
    int thr_lds_len = 0;
    bool dirty = true, ok = false;
    std::string why;
};

static V2State* state(ugvc_ctx* ctx) {
    if (!ctx->v2) ctx->v2 = new V2State();
    return static_cast<V2State*>(ctx->v2);
}

void v2_destroy(ugvc_ctx* ctx) {
    if (!ctx->v2) return;
    V2State* s = static_cast<V2State*>(ctx->v2);
    DeviceBuf* bufs[] = {&s->desc, &s->desc3, &s->lut, &s->thr, &s->css, &s->brackets, &s->brackets3, &s->counters, &s->prof,
                         &s->snp_idx, &s->indel_idx, &s->counters5, &s->eyt, &s->gcr};
    for (auto* b : bufs) if (b->p) dev_free(b->p);
    for (auto& r : s->rec) if (r.p) dev_free(r.p);
    for (auto& r : s->rec5) if (r.p) dev_free(r.p);
    for (auto& g : s->g)
        for (DeviceBuf* b : {&g.d_nodes, &g.d_leaf_idx, &g.d_pairs, &g.d_leaf_f32, &g.d_plane_desc, &g.d_hi4, &g.d_last4, &g.d_p1, &g.d_hi5, &g.d_last5, &g.d_roots5}) if (b->p) dev_free(b->p);
    delete s;
    ctx->v2 = nullptr;
}

static void release_group(V2Group& g) {          // device tables of the previous model of this group, then a clean slate
    for (DeviceBuf* b : {&g.d_nodes, &g.d_leaf_idx, &g.d_pairs, &g.d_leaf_f32, &g.d_plane_desc, &g.d_hi4, &g.d_last4, &g.d_p1, &g.d_hi5, &g.d_last5, &g.d_roots5})
        if (b->p) dev_free(b->p);
    g = V2Group();
}

int clear_model_group(ugvc_ctx* ctx, int gi) {
    V2State* s = state(ctx);
    release_group(s->g[gi]);
    s->dirty = true;
    return 0;
}

static bool search_only_feature(int j) { return j == 0 || j == 1 || j == 5 || j == 13; }  // qual sor vaf gc

int pack_model_group(ugvc_ctx* ctx, int gi, const int32_t* feature, const float* threshold, const int32_t* left,
                     const int32_t* right, int n_nodes, const int32_t* tree_root, int n_trees,
                     const double* leaf_value, int n_leaves, int n_features, int kind, float base, int depth) {
    V2State* s = state(ctx);
    V2Group& g = s->g[gi];
    release_group(g);
    g.set = true;
    g.kind = kind; g.T = n_trees; g.D = depth; g.base = base; g.n_features = n_features;
    g.feature.assign(feature, feature + n_nodes);
    g.threshold.assign(threshold, threshold + n_nodes);
    g.left.assign(left, left + n_nodes);
    g.right.assign(right, right + n_nodes);
    g.roots.assign(tree_root, tree_root + n_trees);
    g.leaf_value.assign(leaf_value, leaf_value + 2 * (size_t)n_leaves);
    g.uthr.assign(kMaxFeatures, {});
    for (int i = 0; i < n_nodes; ++i)
        if (feature[i] >= 0) g.uthr[feature[i]].push_back(threshold[i]);
    for (auto& v : g.uthr) {
        std::sort(v.begin(), v.end());
        v.erase(std::unique(v.begin(), v.end()), v.end());
    }
    s->dirty = true;
    return 0;
}

static void fill_dense(V2Group& g, int t, int src, int level, int idx, std::vector<int>& slot_leaf) {
    // idx: 1-based heap index (root 1, children 2i and 2i+1); level = floor(log2(idx))
    const int D = g.D, NL = 1 << D;
    if (g.feature[src] < 0) {                       // leaf: every slot below it carries its payload
        const int p = idx - (1 << level);
        const int lo = p << (D - level), hi = (p + 1) << (D - level);
        for (int sl = lo; sl < hi; ++sl) slot_leaf[(size_t)t * NL + sl] = g.left[src];
        return;
    }
    const int f = g.feature[src];
    const auto& u = g.uthr[f];
    const int rank = (int)(std::lower_bound(u.begin(), u.end(), g.threshold[src]) - u.begin());
    g.nodes[(size_t)t * NL + idx] = (uint32_t)rank | ((uint32_t)(g.plane[f] * 128) << 16);
    fill_dense(g, t, g.left[src], level + 1, 2 * idx, slot_leaf);
    fill_dense(g, t, g.right[src], level + 1, 2 * idx + 1, slot_leaf);
}

static bool pack_group(V2Group& g) {
    g.ok = false;
    if (g.D < 1 || g.D > 10) { g.why = "tree depth outside 1..10"; return false; }
    // code widths and first-fit-decreasing placement into three 32-bit words
    std::vector<int> order;
    for (int f = 0; f < kMaxFeatures; ++f) {
        const int m = (int)g.uthr[f].size();
        if (m > 4094) { g.why = "more than 4094 distinct thresholds on one feature"; return false; }
        g.width[f] = m ? 32 - __builtin_clz((unsigned)m) : 0;
        g.dword[f] = g.bit_off[f] = g.plane[f] = 0;
        if (m) order.push_back(f);
    }
    std::sort(order.begin(), order.end(), [&](int a, int b) { return g.width[a] > g.width[b]; });
    if (g.layout_done) {
        for (int f = 0; f < kMaxFeatures; ++f) { g.dword[f] = g.jdword[f]; g.bit_off[f] = g.jbit[f]; }
    } else {
        int used[3] = {0, 0, 0};
        for (int f : order) {
            int d = 0;
            while (d < 3 && used[d] + g.width[f] > 32) ++d;
            if (d == 3) { g.why = "rank codes need more than 96 bits"; return false; }
            g.dword[f] = d;
            g.bit_off[f] = used[d];
            used[d] += g.width[f];
        }
    }
    g.n_planes = 0;
    g.plane_desc.clear();
    for (int f = 0; f < kMaxFeatures; ++f)
        if (g.width[f]) {
            g.plane[f] = g.n_planes++;
            g.plane_desc.push_back((uint32_t)g.dword[f] | ((uint32_t)g.bit_off[f] << 2) | ((uint32_t)g.width[f] << 7));
        }
    if (g.n_planes == 0) { g.plane_desc.push_back(0); g.n_planes = 1; }   // stump forest: one dummy plane
    const int D = g.D, T = g.T;
    const size_t n_slot = (size_t)T << D, n_int = n_slot;
    g.nodes.assign(n_int, 0xFFFFu);                 // padded node: rank 65535 on plane 0 -> always left
    std::vector<int> slot_leaf(n_slot, 0);
    for (int t = 0; t < T; ++t) fill_dense(g, t, g.roots[t], 0, 1, slot_leaf);
    size_t lds = n_int * 4 + (size_t)8 * g.n_planes * 128;   // at least 8 waves of code planes must fit
    if (g.kind == UGVC_MODEL_RF) {
        std::map<std::pair<double, double>, int> uniq;
        g.leaf_idx.resize(n_slot);
        g.pairs.clear();
        for (size_t i = 0; i < n_slot; ++i) {
            const std::pair<double, double> key(g.leaf_value[2 * (size_t)slot_leaf[i]], g.leaf_value[2 * (size_t)slot_leaf[i] + 1]);
            auto it = uniq.find(key);
            if (it == uniq.end()) {
                if (uniq.size() >= 65535) { g.why = "more than 65535 distinct leaf payloads"; return false; }
                it = uniq.emplace(key, (int)uniq.size()).first;
                g.pairs.push_back(key.first);
                g.pairs.push_back(key.second);
            }
            g.leaf_idx[i] = (uint16_t)it->second;
        }
        g.n_pairs = (int)uniq.size();
        lds += n_slot * 2 + (size_t)g.n_pairs * 16;
        // single-sum layout
        const size_t H = (size_t)1 << (D - 1);
        double dev = 0.0;
        g.p1.resize((size_t)g.n_pairs);
        for (int k = 0; k < g.n_pairs; ++k) {
            g.p1[k] = g.pairs[2 * (size_t)k + 1];
            const double e = std::fabs(g.pairs[2 * (size_t)k] + g.pairs[2 * (size_t)k + 1] - 1.0);
            dev = !(e <= dev) ? e : dev;                 // NaN payloads disable the layout
            if (!(e == e)) dev = 1.0;
        }
        g.fast4 = dev <= 1e-9 && T <= 4096;
        // a1 and a0 are T-term f64 sums (error <= T^2 * 2^-53 each) of payloads with p0 + p1 = 1 +- dev:
        // a1 - a0 = 2 (a1 - T/2) +- (T dev + 2 T^2 2^-53), and the quotients by T keep a strict order once the
        // difference exceeds a few ulps of T.  Outside this band a1 > T/2 decides; inside, both sums are formed.
        g.band4 = 2.0 * T * dev + 1e-15 * (double)T * (double)T + 1e-14;
        g.hi4.assign((size_t)T * H, 0xFFFFu);
        g.last4.assign((size_t)T * H * 2, 0u);
        for (int t = 0; t < T; ++t) {
            for (size_t i = 1; i < H; ++i) g.hi4[(size_t)t * H + i] = g.nodes[((size_t)t << D) + i];
            for (size_t e = 0; e < H; ++e) {
                g.last4[2 * ((size_t)t * H + e)] = g.nodes[((size_t)t << D) + H + e];
                g.last4[2 * ((size_t)t * H + e) + 1] = (uint32_t)g.leaf_idx[((size_t)t << D) + 2 * e] |
                                                      ((uint32_t)g.leaf_idx[((size_t)t << D) + 2 * e + 1] << 16);
            }
        }
    } else {
        g.leaf_f32.resize(n_slot);
        for (size_t i = 0; i < n_slot; ++i) g.leaf_f32[i] = (float)g.leaf_value[2 * (size_t)slot_leaf[i]];
        lds += n_slot * 4;
    }
    if (lds > (size_t)kLdsBudget) { g.why = "forest does not fit the LDS budget"; return false; }
    g.ok = true;
    return true;
}

// ---- v5 node tables: raw 16-bit codes for the integer-valued features (ugvc_v2.hpp) -----------
static void fill_dense5(const V2Group& g, std::vector<uint32_t>& nodes5, int t, int src, int idx) {
    if (g.feature[src] < 0) return;
    const int f = g.feature[src];
    uint32_t rank;
    if (search_only_feature(f)) {
        const auto& u = g.uthr[f];
        rank = (uint32_t)(std::lower_bound(u.begin(), u.end(), g.threshold[src]) - u.begin());
    } else {
        rank = (uint32_t)std::floor((double)g.threshold[src]) + 1u;      // code > rank <=> x > thr for integer x >= 0
    }
    nodes5[((size_t)t << g.D) + idx] = rank | ((uint32_t)(f * 128) << 16);
    fill_dense5(g, nodes5, t, g.left[src], 2 * idx);
    fill_dense5(g, nodes5, t, g.right[src], 2 * idx + 1);
}

static bool pack_group5(V2Group& g) {
    g.ok5 = false;
    g.used5 = 0;
    for (int f = 0; f < kMaxFeatures; ++f) g.cap5[f] = 0;
    if (!g.ok || g.kind != UGVC_MODEL_RF || !g.fast4) return false;
    for (int f = 0; f < kMaxFeatures; ++f) {
        if (g.uthr[f].empty()) continue;
        g.used5 |= 1u << f;
        if (search_only_feature(f)) continue;
        const double lo = g.uthr[f].front(), hi = g.uthr[f].back();
        if (!(lo >= 0.0) || !(hi < 65533.0)) return false;      // negative / huge / NaN thresholds: the rank-table paths
        g.cap5[f] = (int)std::floor(hi) + 1;
    }
    const int D = g.D, T = g.T;
    const size_t H = (size_t)1 << (D - 1);
    std::vector<uint32_t> nodes5((size_t)T << D, 0xFFFFu);      // padded node: rank 65535 on plane 0 -> always left
    for (int t = 0; t < T; ++t) fill_dense5(g, nodes5, t, g.roots[t], 1);
    // Round 4: ONE heap over all trees of the group.  Node (tree t, level d, position j in the level) sits at dword
    // I = (T + t) 2^d + j - the T roots are the nodes T .. 2T-1 of a heap whose top is virtual - so a child is 2 I + c for
    // EVERY tree: the walk carries I alone, one scalar base serves all trees in flight (round 3: one per tree, sixteen
    // SGPRs of a kernel that spills them), and the roots are consecutive dwords a scalar load brings in (roots5, padded:
    // a batch of 16 trees is one 64-byte s_load).  Levels 0 .. D-2 live in hi5[T H] (entries below T unused), level D-1 in
    // last5 as {node word, left payload | right payload << 16}, entry I - T H = t H + j.  A payload is the BYTE offset
    // 8 * (index of the leaf's class-1 probability in p1): p1 is the first table of the LDS image, so the payload is the
    // address of the final 8-byte read (the instruction's offset field carries the table's base) - one vector instruction
    // less per tree; hence n_pairs <= 8192 (any forest that fits the LDS has fewer: 12 T H + 8 n_pairs bytes).
    if (g.n_pairs > 8192) return false;
    g.hi5.assign((size_t)T * H, 0xFFFFu);
    g.last5.assign((size_t)T * H * 2, 0u);
    g.roots5.assign(((size_t)T + 15) / 16 * 16 + 16, 0xFFFFu);
    for (int t = 0; t < T; ++t) {
        for (int d = 0; d + 1 < D; ++d)
            for (size_t j = 0; j < ((size_t)1 << d); ++j)
                g.hi5[(((size_t)T + t) << d) + j] = nodes5[((size_t)t << D) + ((size_t)1 << d) + j];
        g.roots5[(size_t)t] = nodes5[((size_t)t << D) + 1];
        for (size_t e = 0; e < H; ++e) {
            g.last5[2 * ((size_t)t * H + e)] = nodes5[((size_t)t << D) + H + e];
            g.last5[2 * ((size_t)t * H + e) + 1] = (uint32_t)g.leaf_idx[((size_t)t << D) + 2 * e] * 8u |
                                                  (((uint32_t)g.leaf_idx[((size_t)t << D) + 2 * e + 1] * 8u) << 16);
        }
    }
    g.hi5.resize((g.hi5.size() + 3) & ~(size_t)3, 0xFFFFu);
    g.last5.resize((g.last5.size() + 3) & ~(size_t)3, 0u);
    g.ok5 = true;
    return true;
}

static int count_code(const V2Group& g, int f, float v) {   // rank code of value v for feature f
    const auto& u = g.uthr[f];
    if (g.kind == UGVC_MODEL_RF) return (int)(std::lower_bound(u.begin(), u.end(), v) - u.begin());
    return (int)(std::upper_bound(u.begin(), u.end(), v) - u.begin());
}

static bool boolean_feature(int f) { return f >= 15; }   // inside_hmer_run, close_to_hmer_run, track0..4

// Group-uniform layout used by the v3 kernel: every feature sits in the SAME dword for all
// variant-type groups (bit offsets stay per group), so the packing code selects the accumulator
// with wave-uniform masks; the seven 0/1 features occupy fixed bits 25..31 of dword 2 and are
// packed with one AND (their rank code equals their value when the single threshold lies in
// [0, 1); a constant code 0 when it lies above).  Returns false when the models do not allow it
// (then every group packs on its own and the scoring pass uses the v2 kernel).
static bool joint_layout(V2State* s) {
    int width[UGVC_N_GROUPS][kMaxFeatures] = {};
    int maxw[kMaxFeatures] = {};
    for (int gi = 0; gi < UGVC_N_GROUPS; ++gi) {
        V2Group& g = s->g[gi];
        g.layout_done = false;
        s->boolmask3[gi] = 0;
        if (!g.set) continue;
        for (int f = 0; f < kMaxFeatures; ++f) {
            const int m = (int)g.uthr[f].size();
            if (m > 4094) return false;
            width[gi][f] = m ? 32 - __builtin_clz((unsigned)m) : 0;
            maxw[f] = std::max(maxw[f], width[gi][f]);
            if (boolean_feature(f) && m) {
                if (m > 1 || g.uthr[f][0] < 0.0f) return false;
                if (g.uthr[f][0] < 1.0f) s->boolmask3[gi] |= 1u << (f - 15);
            }
        }
    }
    std::vector<int> order;
    for (int f = 0; f < 15; ++f) if (maxw[f]) order.push_back(f);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return maxw[a] > maxw[b]; });
    const int cap[3] = {32, 32, 25};
    int used[UGVC_N_GROUPS][3] = {};
    int dw[kMaxFeatures] = {}, bit[UGVC_N_GROUPS][kMaxFeatures] = {};
    for (int f : order) {
        int d = 0;
        for (; d < 3; ++d) {
            bool fits = true;
            for (int gi = 0; gi < UGVC_N_GROUPS; ++gi) fits &= used[gi][d] + width[gi][f] <= cap[d];
            if (fits) break;
        }
        if (d == 3) return false;
        dw[f] = d;
        for (int gi = 0; gi < UGVC_N_GROUPS; ++gi) { bit[gi][f] = used[gi][d]; used[gi][d] += width[gi][f]; }
    }
    for (int f = 15; f < kMaxFeatures; ++f) {
        dw[f] = 2;
        for (int gi = 0; gi < UGVC_N_GROUPS; ++gi) bit[gi][f] = 25 + (f - 15);
    }
    for (int gi = 0; gi < UGVC_N_GROUPS; ++gi) {
        V2Group& g = s->g[gi];
        if (!g.set) continue;
        for (int f = 0; f < kMaxFeatures; ++f) { g.jdword[f] = dw[f]; g.jbit[f] = bit[gi][f]; }
        g.layout_done = true;
    }
    for (int f = 0; f < kMaxFeatures; ++f) s->dw3[f] = dw[f];
    return true;
}

int finalize_pack(ugvc_ctx* ctx) {
    V2State* s = state(ctx);
    if (!s->dirty) return 0;
    s->ok = false;
    s->why.clear();
    std::vector<FeatDesc> desc((size_t)UGVC_N_GROUPS * kMaxFeatures, FeatDesc{0, 0, 0, 0});
    std::vector<uint2> desc3((size_t)UGVC_N_GROUPS * kMaxFeatures, make_uint2(0u, 1u));   // unused: dummy LUT entry 0
    std::vector<uint16_t> lut(1, 0);              // entry 0: the code of every feature a group's model never tests
    std::vector<float> thr;
    bool all_ok = true;
    int thr_bits4[4] = {0, 0, 0, 0};
    s->uniform_layout = joint_layout(s);
    for (int gi = 0; gi < UGVC_N_GROUPS; ++gi)
        for (int f = 0; f < kMaxFeatures; ++f)
            if (search_only_feature(f)) desc3[(size_t)gi * kMaxFeatures + f] = make_uint2(0u, 0u);   // empty threshold slice
    for (auto& g : s->g)
        if (g.set && !pack_group(g)) { all_ok = false; s->why = g.why; }
    for (auto& g : s->g)
        if (g.set) pack_group5(g);
    if (all_ok) {
        // threshold table: search-only features of every group first (that prefix is staged in LDS)
        s->thr0_len = s->thr_lds_len = 0;
        for (int pass = 0; pass < 2; ++pass) {
            for (int gi = 0; gi < UGVC_N_GROUPS; ++gi) {
                V2Group& g = s->g[gi];
                if (pass == 0 && gi == 1) s->thr0_len = (int)thr.size();      // group 0's float slices lead the table
                if (!g.set) continue;
                for (int f = 0; f < kMaxFeatures; ++f) {
                    if (g.uthr[f].empty() || search_only_feature(f) != (pass == 0)) continue;
                    FeatDesc& d = desc[(size_t)gi * kMaxFeatures + f];
                    d.thr = (uint32_t)thr.size() | ((uint32_t)g.uthr[f].size() << 20);   // off 20 bits | len 12 bits
                    thr.insert(thr.end(), g.uthr[f].begin(), g.uthr[f].end());
                    d.pack = (uint32_t)g.dword[f] | ((uint32_t)g.bit_off[f] << 2) | ((uint32_t)g.width[f] << 7);
                    uint2& d3 = desc3[(size_t)gi * kMaxFeatures + f];
                    const uint32_t pk3 = ((uint32_t)g.dword[f] << 16) | ((uint32_t)g.bit_off[f] << 18);
                    if (pass == 0) {
                        d.lut = 2u << 30;
                        d3 = make_uint2((d.thr & 0xFFFFFu) | (2u << 30), (uint32_t)g.uthr[f].size() | pk3);
                        const int m = (int)g.uthr[f].size();
                        const int k4 = f == 0 ? 0 : (f == 1 ? 1 : (f == 5 ? 2 : 3));
                        thr_bits4[k4] = std::max(thr_bits4[k4], m > 0 ? 32 - __builtin_clz((unsigned)m) : 0);
                    } else {
                        const double top = std::floor((double)g.uthr[f].back());
                        const int len = (int)std::min(8192.0, std::max(1.0, top + 2.0));
                        d.lut = (uint32_t)lut.size() | (1u << 30);
                        d.lut_len = (uint32_t)len;
                        for (int v = 0; v < len; ++v) lut.push_back((uint16_t)count_code(g, f, (float)v));
                        // kind 3: the table reaches one past the top threshold, so every larger value takes its last
                        // entry (no threshold search); kind 1: table cut at 8192 entries, larger values search
                        d3 = make_uint2((d.lut & 0xFFFFFu) | ((top + 2.0 <= 8192.0 ? 3u : 1u) << 30), (uint32_t)len | pk3);
                    }
                }
            }
            // (a group without a model must not leave the previous configuration's length behind)
            if (pass == 0) s->thr_lds_len = (int)thr.size();
        }
        if (thr.size() >= (1u << 20) || lut.size() >= (1u << 20)) { all_ok = false; s->why = "code tables too large"; }
    }
    if (all_ok) {
        UGVC_HIP(hipSetDevice(ctx->device));
        if (upload(ctx, s->desc, desc.data(), desc.size() * sizeof(FeatDesc))) return -1;
        if (upload(ctx, s->desc3, desc3.data(), desc3.size() * sizeof(uint2))) return -1;
        for (int k = 0; k < 4; ++k) s->thr_bits4[k] = thr_bits4[k];
        s->h_desc3 = desc3;
        {
            // v5, group 0: the thresholds of qual / sor / vaf as complete binary search trees in level order (index 1 =
            // root, children 2 i and 2 i + 1, padded with +inf): a descent is `i = 2 i + (t < x)`, its reads at one level
            // fall on consecutive LDS words (no power-of-two bank pile-up), and the leaf index is the rank
            std::vector<float> eyt;
            const int fj[3] = {0, 1, 5};
            for (int k = 0; k < 3; ++k) {
                const std::vector<float> empty;
                const auto& u = s->g[0].set ? s->g[0].uthr[fj[k]] : empty;
                const int n = (int)u.size();
                int bits = 0;
                while ((1 << bits) - 1 < n) ++bits;
                s->eyt_bits[k] = bits;
                s->eyt_off[k] = (int)eyt.size();
                const int size = 1 << bits;
                std::vector<float> e((size_t)size, std::numeric_limits<float>::infinity());
                int pos = 0;
                std::vector<std::pair<int, int>> stack;       // in-order walk of the implicit tree
                int i = 1;
                while (i < size || !stack.empty()) {
                    while (i < size) { stack.push_back({i, 0}); i = 2 * i; }
                    i = stack.back().first;
                    stack.pop_back();
                    e[(size_t)i] = pos < n ? u[(size_t)pos] : std::numeric_limits<float>::infinity();
                    ++pos;
                    i = 2 * i + 1;
                }
                eyt.insert(eyt.end(), e.begin(), e.end());
            }
            eyt.resize((eyt.size() + 3) & ~(size_t)3, std::numeric_limits<float>::infinity());
            s->eyt_len = (int)eyt.size();
            if (upload(ctx, s->eyt, eyt.data(), eyt.size() * 4)) return -1;
            // v5: gc_content takes 121 values (count / len with len, count in 0..10): its rank among a group's thresholds
            // is a table [group][len * 11 + count] (the fused kernel used to build it in its prologue: 363 threads looping
            // over the thresholds, ~10 us per launch)
            std::vector<uint16_t> gcr(384 * 1, 0);
            for (int g = 0; g < UGVC_N_GROUPS; ++g) {
                const std::vector<float> empty;
                const auto& u = s->g[g].set ? s->g[g].uthr[13] : empty;
                for (int r = 0; r < 121; ++r) {
                    const int len = r / 11, cnt = r % 11;
                    const float f = (len > 0 && cnt <= len) ? (float)((double)cnt / (double)len) : 0.0f;
                    int rank = 0;
                    for (float t : u) rank += t < f ? 1 : 0;
                    gcr[(size_t)g * 121 + r] = (uint16_t)rank;
                }
            }
            if (upload(ctx, s->gcr, gcr.data(), gcr.size() * 2)) return -1;
        }
        for (int k = 0; k < 4; ++k) {
            const int f = k == 0 ? 0 : (k == 1 ? 1 : (k == 2 ? 5 : 13));
            const int m = s->g[0].set ? (int)s->g[0].uthr[f].size() : 0;
            s->thr0_bits4[k] = m > 0 ? 32 - __builtin_clz((unsigned)m) : 0;
        }
        if (upload(ctx, s->lut, lut.data(), lut.size() * 2)) return -1;
        thr.resize((thr.size() + 3) & ~(size_t)3, 0.f);          // K1 copies the LDS-resident prefix as float4
        if (upload(ctx, s->thr, thr.data(), thr.size() * 4)) return -1;
        for (auto& g : s->g) {
            if (!g.set) continue;
            if (upload(ctx, g.d_nodes, g.nodes.data(), g.nodes.size() * 4)) return -1;
            if (upload(ctx, g.d_plane_desc, g.plane_desc.data(), g.plane_desc.size() * 4)) return -1;
            if (g.kind == UGVC_MODEL_RF) {
                if (upload(ctx, g.d_leaf_idx, g.leaf_idx.data(), g.leaf_idx.size() * 2)) return -1;
                if (upload(ctx, g.d_pairs, g.pairs.data(), g.pairs.size() * 8)) return -1;
                // padded to whole 16-byte pieces: the forest kernel fills its LDS with 16-byte loads
                g.hi4.resize((g.hi4.size() + 3) & ~(size_t)3, 0xFFFFu);
                g.last4.resize((g.last4.size() + 3) & ~(size_t)3, 0u);
                g.p1.resize((g.p1.size() + 1) & ~(size_t)1, 0.0);
                if (upload(ctx, g.d_hi4, g.hi4.data(), g.hi4.size() * 4)) return -1;
                if (upload(ctx, g.d_last4, g.last4.data(), g.last4.size() * 4)) return -1;
                if (upload(ctx, g.d_p1, g.p1.data(), g.p1.size() * 8)) return -1;
                if (g.ok5) {
                    if (upload(ctx, g.d_hi5, g.hi5.data(), g.hi5.size() * 4)) return -1;
                    if (upload(ctx, g.d_last5, g.last5.data(), g.last5.size() * 4)) return -1;
                    if (upload(ctx, g.d_roots5, g.roots5.data(), g.roots5.size() * 4)) return -1;
                }
            } else if (upload(ctx, g.d_leaf_f32, g.leaf_f32.data(), g.leaf_f32.size() * 4)) return -1;
        }
        UGVC_HIP(hipStreamSynchronize(ctx->stream));
        s->ok = true;
    }
    s->dirty = false;
    return 0;
}

// ---- cycle-skip LUT ------------------------------------------------------------------------
// For a single-base substitution the status depends only on (last left base, ref, alt, first
// right base): the flow that consumes the left neighbour is the same for both alleles, and
// both key streams re-synchronise on the right neighbour.  tests/test_host_logic.py checks the
// table against the full 11-mer flow-key computation of the oracle on exhaustive contexts.
static int flow_key(const int* seq, int n, const uint8_t flow[4], int* key) {
    int p = 0, s = 0;
    while (p < n) {
        const int b = flow[s & 3];
        int h = 0;
        while (p + h < n && seq[p + h] == b) ++h;
        key[s++] = h;
        p += h;
    }
    return s;
}

void host_css_lut(const uint8_t flow[4], uint8_t out[256]) {
    for (int l1 = 0; l1 < 4; ++l1) for (int r = 0; r < 4; ++r) for (int a = 0; a < 4; ++a) for (int r1 = 0; r1 < 4; ++r1) {
        int sr[3] = {l1 + 1, r + 1, r1 + 1}, sa[3] = {l1 + 1, a + 1, r1 + 1};
        int kr[16], ka[16];
        const int nr = flow_key(sr, 3, flow, kr), na = flow_key(sa, 3, flow, ka);
        int st = 0;
        if (nr != na) st = 2;
        else
            for (int i = 0; i < nr; ++i)
                if (kr[i] != ka[i] && (kr[i] == 0 || ka[i] == 0)) st = 1;
        out[(l1 << 6) | (r << 4) | (a << 2) | r1] = (uint8_t)st;
    }
}

int build_css_lut(ugvc_ctx* ctx) {
    V2State* s = state(ctx);
    uint8_t t[256];
    host_css_lut(ctx->flow, t);
    UGVC_HIP(hipSetDevice(ctx->device));
    if (upload(ctx, s->css, t, 256)) return -1;
    UGVC_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

bool v2_available(ugvc_ctx* ctx) {
    V2State* s = state(ctx);
    if (finalize_pack(ctx)) return false;
    bool any = false;
    for (auto& g : s->g) any |= g.set;
    return s->ok && any;
}

const char* v2_reason(ugvc_ctx* ctx) { return state(ctx)->why.c_str(); }

int v2_phase_clocks(ugvc_ctx* ctx, uint64_t out[8], int reset) {
    V2State* s = state(ctx);
    UGVC_HIP(hipSetDevice(ctx->device));
    if (ensure(s->prof, 64)) return -1;
    UGVC_HIP(hipStreamSynchronize(ctx->stream));
    UGVC_HIP(hipMemcpy(out, s->prof.p, 64, hipMemcpyDeviceToHost));
    if (reset) UGVC_HIP(hipMemset(s->prof.p, 0, 64));
    return 0;
}

int v2_fill_args(ugvc_ctx* ctx, V2Args& v, int64_t n, int n_tiles) {
    V2State* s = state(ctx);
    if (!s->css.p && build_css_lut(ctx)) return -1;
    const int n_blocks = n_tiles > 0 ? n_tiles : (int)((n + kBlock - 1) / kBlock);
    const size_t shard_cap = (size_t)((n_blocks + kShards - 1) / kShards) * kBlock;
    v.shard_cap = (int)shard_cap;
    for (int gi = 0; gi < UGVC_N_GROUPS; ++gi) {
        const V2Group& g = s->g[gi];
        PackedGroupView& p = v.pg[gi];
        p = PackedGroupView{};
        if (!g.set) continue;
        p.ok = 1; p.kind = g.kind; p.T = g.T; p.D = g.D; p.base = g.base; p.n_pairs = g.n_pairs;
        p.n_planes = g.n_planes;
        p.plane_desc = g.d_plane_desc.as<uint32_t>();
        p.nodes = g.d_nodes.as<uint32_t>();
        p.leaf_idx = g.d_leaf_idx.as<uint16_t>();
        p.pairs = g.d_pairs.as<double2>();
        p.leaf_f32 = g.d_leaf_f32.as<float>();
        p.fast4 = g.kind == UGVC_MODEL_RF && g.fast4;
        p.band = g.band4;
        p.inv_T = g.T > 0 ? 1.0 / (double)g.T : 0.0;
        p.hi4 = g.d_hi4.as<uint32_t>();
        p.last4 = g.d_last4.as<uint2>();
        p.p1 = g.d_p1.as<double>();
        if (ensure(s->rec[gi], (size_t)kShards * shard_cap * 16)) return -1;
        v.records[gi] = s->rec[gi].as<uint4>();
    }
    v.desc = s->desc.as<FeatDesc>();
    v.lut = s->lut.as<uint16_t>();
    v.thr = s->thr.as<float>();
    v.thr_lds_len = std::min(s->thr_lds_len, 4096);   // kThrLds floats are staged in LDS; the rest is read from L2
    v.css_lut = s->css.as<uint8_t>();
    v.n_blocks = n_blocks;
    if (ensure(s->brackets, (size_t)(v.n_blocks + 1) * kJoinArrays * 4)) return -1;
    if (ensure(s->counters, (size_t)UGVC_N_GROUPS * kShards * kCounterStride * 4)) return -1;
    v.brackets = s->brackets.as<int32_t>();
    v.counters = s->counters.as<uint32_t>();
    // v3
    if (ensure(s->brackets3, (size_t)(v.n_blocks + 1) * 8 * 4 + 64)) return -1;
    v.brackets3 = s->brackets3.as<int32_t>();
    v.desc3 = s->desc3.as<uint2>();
    for (int k = 0; k < 4; ++k) v.thr_bits4[k] = s->thr_bits4[k];
    for (int f = 0; f < kMaxFeatures; ++f) v.dw3[f] = (uint8_t)s->dw3[f];
    for (int g = 0; g < UGVC_N_GROUPS; ++g) v.boolmask3[g] = s->boolmask3[g];
    for (int t = 0; t < 8; ++t) v.na3[t] = 0;
    v.na3[0] = ctx->has_runs ? (int)ctx->runs_n : 0;
    for (int t = 0; t < ctx->n_tracks; ++t) v.na3[1 + t] = (int)ctx->trk_n[t];
    v.na3[kJoin3 - 1] = (int)ctx->n_bl;
    if (ensure(s->prof, 64)) return -1;
    v.prof = s->prof.as<unsigned long long>();
    return 0;
}

// Dense depth-6 tables of one group's additive ensemble for the leaf-matrix GEMM / row traversal
// kernels (kernels_gemm.hip): nodes[t][heap index - 1] = {threshold, feature}, slot 63 padding;
// leaves[t][0..63].  A leaf above level 6 owns every slot below it; the padded decisions on the way
// (threshold +inf, feature 0) are immaterial because both sides carry the same payload.
static void gemm_fill(const V2Group& g, int t, int src, int level, int idx, std::vector<float2>& nodes, std::vector<float>& leaves) {
    if (g.feature[src] < 0) {
        const int p = idx - (1 << level);
        for (int sl = p << (6 - level); sl < (p + 1) << (6 - level); ++sl)
            leaves[(size_t)t * 64 + sl] = (float)g.leaf_value[2 * (size_t)g.left[src]];
        return;
    }
    float2 nd;
    nd.x = g.threshold[src];
    nd.y = __builtin_bit_cast(float, (int)g.feature[src]);
    nodes[(size_t)t * 64 + idx - 1] = nd;
    gemm_fill(g, t, g.left[src], level + 1, 2 * idx, nodes, leaves);
    gemm_fill(g, t, g.right[src], level + 1, 2 * idx + 1, nodes, leaves);
}

int gemm_model(ugvc_ctx* ctx, int group, std::vector<float2>& nodes, std::vector<float>& leaves, int& T, int& kind, float& base) {
    const V2Group& g = state(ctx)->g[group];
    if (!g.set) return fail("no model uploaded for group " + std::to_string(group));
    if (g.kind != UGVC_MODEL_GBT) return fail("the leaf-matrix GEMM formulation is implemented for additive (XGBoost-style) ensembles");
    if (g.D > 6) return fail("the leaf-matrix GEMM formulation needs trees of depth <= 6 (63 x 64 path matrix)");
    T = g.T; kind = g.kind; base = g.base;
    float2 pad;
    pad.x = std::numeric_limits<float>::infinity();
    pad.y = __builtin_bit_cast(float, 0);
    nodes.assign((size_t)T * 64, pad);
    leaves.assign((size_t)T * 64, 0.f);
    for (int t = 0; t < T; ++t) gemm_fill(g, t, g.roots[t], 0, 1, nodes, leaves);
    return 0;
}

int v5_fused_waves(const V5Args& v);

// Everything the v5 launches need for the resident variants: node tables, index lists, brackets, record lists,
// and the LDS layout of a wave's staged side-table slices (capacities follow the tables' densities).
int v5_fill_args(ugvc_ctx* ctx, V5Args& v, const FilterArgs& a, bool scoring, int wg_per_cu) {
    V2State* s = state(ctx);
    if (!s->css.p && build_css_lut(ctx)) return -1;
    v = V5Args{};
    v.f = a;
    const int64_t n = a.n;
    for (int gi = 0; gi < UGVC_N_GROUPS; ++gi) {
        const V2Group& g = s->g[gi];
        PackedGroupView& p = v.pg[gi];
        p = PackedGroupView{};
        v.used5[gi] = 0;
        for (int f = 0; f < kMaxFeatures; ++f) v.cap5[gi][f] = 0;
        // (a feature-matrix launch - scoring == false - ranks and walks nothing: no forest, no threshold table in its LDS
        // budget, so a model the scoring pass would hand to v3 for size cannot make it fail)
        if (!g.set || !g.ok5 || !scoring) continue;
        p.ok = 1; p.kind = g.kind; p.T = g.T; p.D = g.D; p.base = g.base; p.n_pairs = g.n_pairs;
        p.n_planes = kMaxFeatures;
        p.pairs = g.d_pairs.as<double2>();
        p.fast4 = 1;
        p.band = g.band4;
        p.inv_T = g.T > 0 ? 1.0 / (double)g.T : 0.0;
        p.hi4 = g.d_hi5.as<uint32_t>();
        p.last4 = g.d_last5.as<uint2>();
        p.roots = g.d_roots5.as<uint32_t>();
        p.p1 = g.d_p1.as<double>();
        v.used5[gi] = g.used5;
        for (int f = 0; f < kMaxFeatures; ++f) v.cap5[gi][f] = g.cap5[f];
    }
    v.thr = s->thr.as<float>();
    v.desc3 = s->desc3.as<uint2>();
    v.thr_lds_len = scoring ? s->thr_lds_len : 0;
    v.thr0_len = scoring ? s->thr0_len : 0;
    v.thr_bits = 0;
    for (int gi = 1; gi < UGVC_N_GROUPS; ++gi)
        for (int f : {0, 1, 5}) {
            const int m = s->g[gi].set ? (int)s->g[gi].uthr[f].size() : 0;
            v.thr_bits = std::max(v.thr_bits, m > 0 ? 32 - __builtin_clz((unsigned)m) : 0);
        }
    v.eyt = s->eyt.as<float>();
    v.eyt_len = scoring ? s->eyt_len : 0;
    v.gcr = s->gcr.as<uint16_t>();
    for (int k = 0; k < 3; ++k) { v.eyt_off[k] = s->eyt_off[k]; v.eyt_bits[k] = s->eyt_bits[k]; }
    v.css_lut = s->css.as<uint8_t>();
    // rows per workgroup of the fused kernel: the callset split evenly over the CUs (at least kMinRowsWg5 rows each)
    const int64_t n_slots = (int64_t)ctx->n_cus * std::max(wg_per_cu, 1);
    v.rows_wg = (int)std::max<int64_t>((n + n_slots - 1) / n_slots, kMinRowsWg5);
    const int64_t n_wg = (n + v.rows_wg - 1) / v.rows_wg;
    v.list_stride = (v.rows_wg + kTile5 - 1) / kTile5 * kTile5 + kTile5;
    // record lists: a workgroup's indel tiles go round-robin over the kShards lists of their group
    const int64_t tiles_wg = v.list_stride / kTile5;
    // (a tile is a run of <= 64 list entries from any offset - kernels_v5.hip: tile_cut - and every lane's record goes to the
    // shard of the 64-entry block its entry lies in: at most 64 records per workgroup and block)
    v.shard_cap5 = (int)(n_wg * ((tiles_wg + kShards - 1) / kShards) * kTile5);
    if (ensure(s->snp_idx, (size_t)n_wg * v.list_stride * 4) || ensure(s->indel_idx, (size_t)n_wg * v.list_stride * 4)) return -1;
    const size_t c5_bytes = (size_t)UGVC_N_GROUPS * kShards * kCounterStride * 4;
    {
        const void* before = s->counters5.p;
        if (ensure(s->counters5, 2 * c5_bytes)) return -1;
        if (s->counters5.p != before) s->c5_dirty[0] = s->c5_dirty[1] = true;
    }
    v.snp_idx = s->snp_idx.as<uint32_t>();
    v.indel_idx = s->indel_idx.as<uint32_t>();
    {
        // the forest kernel of a pass zeroes the set the NEXT pass counts into; a set that missed that (the first two
        // passes, a pass without indel models in between) is cleared here
        // (a feature-matrix launch - scoring == false - writes no records and runs no forest kernel: it must leave the
        // two counter sets and what is known about them alone.  Found by the determinism test: a feature-matrix launch
        // between two scoring passes used to mark the set the first pass had counted into as "zeroed by the forest
        // kernel", and the second pass appended its records behind the first one's.)
        const int cur = s->c5_set;
        if (scoring) s->c5_set ^= 1;
        uint8_t* base = static_cast<uint8_t*>(s->counters5.p);
        if (scoring && s->c5_dirty[cur]) {
            UGVC_HIP(hipMemsetAsync(base + cur * c5_bytes, 0, c5_bytes, ctx->stream));
            s->c5_dirty[cur] = false;
        }
        v.counters = reinterpret_cast<uint32_t*>(base + cur * c5_bytes);
        v.counters_next = reinterpret_cast<uint32_t*>(base + (cur ^ 1) * c5_bytes);
        // (a callset without indels writes no records: no forest launch, nothing to zero)
        const bool forest = scoring && ctx->n_indel > 0 && !(a.ablate & 262144) && (v.pg[1].ok || v.pg[2].ok);
        v.run_forest = forest ? 1 : 0;
        if (forest) {
            s->c5_dirty[cur] = true;
            s->c5_dirty[cur ^ 1] = false;
        }
    }
    for (int gi = 1; gi < UGVC_N_GROUPS; ++gi) {
        v.rec5[gi] = nullptr;
        if (!v.pg[gi].ok) continue;
        if (ensure(s->rec5[gi], (size_t)kShards * v.shard_cap5 * kRec5Dwords * 4)) return -1;
        v.rec5[gi] = s->rec5[gi].as<uint4>();
    }
    // staged slice of an interval table in the SNP path: 64 rows, or 128 for the tables whose density puts more than
    // ~40 rows under a tile of 64 substitutions (about 96 consecutive variants at a WGS class mix), densest first,
    // while a wave's scratch has room; a tile that needs more searches HBM.  Blacklist: kBlCap5 keys.
    for (int t = 0; t < kJoin5; ++t) { v.na[t] = 0; v.jcap[t] = 64; v.joff[t] = 0; }
    v.na[0] = ctx->has_runs ? (int)ctx->runs_n : 0;
    for (int t = 0; t < ctx->n_tracks; ++t) v.na[1 + t] = (int)ctx->trk_n[t];
    v.na[kJoin5 - 1] = (int)ctx->n_bl;
    v.jcap[kJoin5 - 1] = kBlCap5;
    {
        const int n_int = 1 + ctx->n_tracks;
        int budget = (3072 - kBlCap5 * 8) / 4 - n_int * 128;                 // dwords left of 3 KB after 64 rows of every table
        std::vector<int> order;
        for (int t = 0; t < n_int; ++t) order.push_back(t);
        std::sort(order.begin(), order.end(), [&](int x, int y) { return v.na[x] > v.na[y]; });
        for (int t : order) {
            const double per_tile = (double)v.na[t] / (double)std::max<int64_t>(ctx->density_n ? ctx->density_n : n, 1) * 96.0;
            if (per_tile > 40.0 && budget >= 128) { v.jcap[t] = 128; budget -= 128; }
        }
        int used = 0;
        for (int t = 0; t < n_int; ++t) { v.joff[t] = used; used += 2 * v.jcap[t]; }
        v.joff[kJoin5 - 1] = used;                                           // 8-byte keys: `used` is even
        used += 2 * kBlCap5;
        v.scratch_bytes = (std::max(used * 4, kMaxFeatures * 128) + 63) & ~63;                 // staged slices, then the code planes
        v.scratch_indel = kTile5 * 13 * 4;                                                      // 48-byte window rows, stride 13 dwords
    }
    // indel tiles span ~64 / (indel share) rows of the callset: a table that puts more than ~100 rows under one gets
    // the six-rows-per-lane slice (kernels_v5.hip: wide_load)
    v.iwide = 0;
    {
        const double tiles = std::max<double>((double)std::max<int64_t>(ctx->n_indel, 1) / kTile5, 1.0);
        for (int t = 0; t < 1 + ctx->n_tracks; ++t)
            if ((double)v.na[t] / tiles > 100.0) v.iwide |= 1u << t;
        if (const char* e = getenv("UGVC_IWIDE")) v.iwide = (uint32_t)atoi(e);        // (profiling)
    }
    // (an indel tile relative to an SNP tile in the wave-role split: 0.85 since round 4's indel tiles - one memory round trip less
    // per tile, one atomic instead of four; at 1.0 the workgroups with the most indel tiles went to four indel waves and were the
    // launch's slowest, profiles/r04_indel_cost_sweep.txt)
    v.indel_w = 218;
    if (const char* e = getenv("UGVC_INDEL_COST")) v.indel_w = std::max(1, (int)(atof(e) * 256.0));      // (profiling)
    for (auto& x : v.snp_cum) x = 0;
    if (const char* e = getenv("UGVC_SNP_W")) {                           // (profiling) "w0,w1,...": tile weights of the SNP waves
        int k = 0;
        for (const char* p = e; *p && k < 16; ++k) {
            v.snp_cum[k + 1] = (unsigned short)(v.snp_cum[k] + atoi(p));
            while (*p && *p != ',') ++p;
            if (*p == ',') ++p;
        }
        for (; k < 16; ++k) v.snp_cum[k + 1] = v.snp_cum[k];
    }
    v.n_indel_waves = 0;
    v.n_waves = v5_fused_waves(v);
    if (v.n_waves == 0) return fail("internal: the SNP forest does not fit the fused kernel's LDS");
    // indel tiles are ~1/5 of a WGS callset's tiles and take about as long as an SNP tile with its walk: 1/4 of the waves
    // (kernel variant bits 24-27 override, profiling)
    v.n_indel_waves = ((a.ablate >> 24) & 15) ? ((a.ablate >> 24) & 15) : v.n_waves / 4;
    if (v.n_indel_waves >= v.n_waves) v.n_indel_waves = v.n_waves - 1;
    if (v5_fused_waves(v) < v.n_waves) v.n_waves = v5_fused_waves(v);   // (the indel waves' larger scratch)
    if (v.n_waves == 0) return fail("internal: the SNP forest does not fit the fused kernel's LDS");
    if (v.n_indel_waves >= v.n_waves) v.n_indel_waves = v.n_waves - 1;
    // indel tiles: all two-rows-per-lane slices (1 KB each) and the blacklist keys (1 KB) side by side in the wave's
    // scratch if the LDS has room for that at the same number of waves - one staging round instead of one per pair
    v.indel_one_round = 0;
    {
        int n_narrow = 0;
        for (int t = 0; t < 1 + ctx->n_tracks; ++t)
            if (!(t == 0 && !ctx->has_runs) && !((v.iwide >> t) & 1)) ++n_narrow;
        const int want = 1024 * (n_narrow + (ctx->n_bl > 0 ? 1 : 0));
        if (want > v.scratch_indel && !getenv("UGVC_INDEL_ROUNDS")) {
            const int keep = v.scratch_indel;
            v.scratch_indel = want;
            if (v5_fused_waves(v) >= v.n_waves) v.indel_one_round = 1;
            else v.scratch_indel = keep;
        } else if (want <= v.scratch_indel && !getenv("UGVC_INDEL_ROUNDS")) v.indel_one_round = 1;
    }
    return 0;
}

bool v5_available(ugvc_ctx* ctx) {
    V2State* s = state(ctx);
    if (!v2_available(ctx)) return false;
    bool any = false;
    for (auto& g : s->g) {
        if (!g.set) continue;
        if (!g.ok5) return false;                             // pair-sum forests, XGBoost-style ensembles, odd thresholds: v3
        any = true;
    }
    if (!any) return false;
    if (s->thr_lds_len > kThr3) return false;
    if (ctx->has_runs && !ctx->runs_fast) return false;
    for (int t = 0; t < ctx->n_tracks; ++t)
        if (!ctx->trk_fast[t]) return false;
    // 32-bit byte offsets into every resident array (ldg32): rows x element size stay below 4 GiB
    if (ctx->n >= ((int64_t)1 << 30) || ctx->n_bl >= ((int64_t)1 << 29)) return false;
    if (ctx->has_runs && ctx->runs_n >= ((int64_t)1 << 30)) return false;
    for (int t = 0; t < ctx->n_tracks; ++t)
        if (ctx->trk_n[t] >= ((int64_t)1 << 30)) return false;
    // LDS of the forest kernel: an indel group's forest + four waves of code planes + its static words
    // (kernels_v5.hip: k5_forest_lds + 1088) - a forest near the v3 budget goes to v3 instead of failing at launch
    for (int gi = 1; gi < UGVC_N_GROUPS; ++gi) {
        const V2Group& g = s->g[gi];
        if (!g.set) continue;
        const size_t n_hi = ((size_t)g.T << g.D) / 2;
        if (n_hi * 12 + (size_t)g.n_pairs * 8 + 48 + 4 * (size_t)kMaxFeatures * 128 + 1088 > 158 * 1024) return false;
    }
    // LDS: group 0's forest + the thresholds + at least 8 waves of scratch
    const V2Group& g0 = s->g[0];
    size_t forest = 0;
    if (g0.set) forest = ((size_t)g0.T << g0.D) / 2 * 12 + (size_t)g0.n_pairs * 8 + 64;
    if (forest + (size_t)(s->thr_lds_len - s->thr0_len) * 33 / 8 + (size_t)s->eyt_len * 4 + 2048 + 6 * 3072 + 2 * 3328 > 158 * 1024) return false;
    return true;
}

// the feature matrix on the fused kernel's featurize waves (kernels_v5.hip, WX): needs what those tiles assume of the
// side tables (sorted, disjoint runs; 32-bit offsets) - no model at all
bool fm5_available(ugvc_ctx* ctx) {
    if (ctx->kernel_variant & 256) return false;              // (bit 8 forces the universal kernel everywhere)
    if (ctx->has_runs && !ctx->runs_fast) return false;
    for (int t = 0; t < ctx->n_tracks; ++t)
        if (!ctx->trk_fast[t]) return false;
    if (ctx->n >= ((int64_t)1 << 30) || ctx->n_bl >= ((int64_t)1 << 29)) return false;
    if (ctx->has_runs && ctx->runs_n >= ((int64_t)1 << 30)) return false;
    for (int t = 0; t < ctx->n_tracks; ++t)
        if (ctx->trk_n[t] >= ((int64_t)1 << 30)) return false;
    const size_t F = (size_t)(UGVC_N_BASE_FEATURES + ctx->n_tracks);
    if ((size_t)ctx->n * F * 4 >= ((size_t)1 << 32)) return false;      // (row offsets are formed in 64 bits, but keep X below the 32-bit list indices' reach)
    return true;
}

bool v3_available(ugvc_ctx* ctx) {
    V2State* s = state(ctx);
    if (!v2_available(ctx)) return false;
    if (s->thr_lds_len > kThr3) return false;                 // float-feature thresholds must fit K1's LDS slice
    if (!s->uniform_layout) return false;
    if (ctx->n_contigs > 256) return false;
    if (ctx->has_runs && !ctx->runs_fast) return false;
    for (int t = 0; t < ctx->n_tracks; ++t)
        if (!ctx->trk_fast[t]) return false;
    return true;
}

}  // namespace ugvc

extern "C" {
// Host-only helper (no GPU needed): the single-base cycle-skip table for a flow order, so the
// CPU tests can check it against the oracle's full flow-key computation.
int ugvc_host_css_lut(const char* flow4, uint8_t out[256]) {
    uint8_t f[4];
    int seen = 0;
    if (!out) return ugvc::fail("out is NULL");
    for (int k = 0; k < 4; ++k) {
        switch (flow4 ? flow4[k] : 0) {
            case 'A': case 'a': f[k] = 1; break;
            case 'C': case 'c': f[k] = 2; break;
            case 'G': case 'g': f[k] = 3; break;
            case 'T': case 't': f[k] = 4; break;
            default: return ugvc::fail("flow order must be a permutation of ACGT");
        }
        seen |= 1 << f[k];
    }
    if (seen != 0x1e || flow4[4] != 0) return ugvc::fail("flow order must be a permutation of ACGT");
    ugvc::host_css_lut(f, out);
    return 0;
}
}
