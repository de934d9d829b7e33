// v3 scoring pass (gfx950): K0 brackets of the searched side arrays, K1 persistent featurize +
// lookup + quantise, K2 LDS-resident forest walk with paired child fetches.
//
// What changed against v2 (kernels_v2.hip) and why - measured on MI355X, 5 M variants:
// v2's K1 ran at 3 waves/SIMD (52 KB LDS) and was bound by chains of dependent LDS/global
// round trips (13 serial staging loops, 13 serial binary searches of ~9 dependent LDS reads,
// 20 serial quantisation searches, 64-bit bounds-checked window accessors); K2 paid two
// dependent LDS gathers per node visit with 2-4 way bank conflicts.
//   K1: * persistent workgroups: thresholds / descriptors / contig table / cycle-skip LUT are
//         loaded into LDS once per workgroup, not once per 256 variants;
//       * all searches are power-of-two descents run in LOCK-STEP over the tables (7 joins,
//         then 4 float features), so their LDS latencies overlap instead of adding up;
//       * `inside interval` needs only the rank among STARTS plus two direct reads of ENDS
//         (tables are validated sorted at upload), halving the searches;
//       * the reference window is 48 B/lane in a conflict-free row-per-lane LDS layout read
//         with byte loads at immediate offsets; contig-edge lanes patch their window once, so
//         the common path has no bounds checks and no 64-bit arithmetic;
//       * allele bytes, contig CSR pointers and LUT codes are fetched as batches of
//         independent loads; one returning atomic per wave and group, issued before the joins
//         and consumed after the quantisation.
//   K2: * VALU-issue bound (rocprofv3 SQ counters: 73 % VALU busy, 62 % LDS busy): a node visit is
//         cut to 4 VALU (address, code address via SDWA, compare via SDWA, v_addc index update);
//         8 independent trees per lane cover the two dependent LDS latencies of a level;
//       * code planes use a lane -> halfword permutation that is bank-conflict free.
// Semantics are those of the oracle (oracle/oracle.py); parity tests run v3, v2 and v1.
#include "ugvc_walk.hpp"

namespace ugvc {


// ---- K0: brackets3[b][a] = first index of searched array a (a < 6: starts of table a; a == 6:
// blacklist keys) that is >= the first variant of tile b; row n_tiles holds the array lengths.
__global__ void bracket3_kernel(const V2Args v) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nb = v.n_blocks;
    // the record-list counters of this pass start at zero (saves a memset launch per pass)
    if (gid < UGVC_N_GROUPS * kShards) v.counters[gid * kCounterStride] = 0;
    const int b = (int)(gid >> 3), a = (int)(gid & 7);
    if (b > nb || a >= kJoin3) return;
    const FilterArgs& f = v.f;
    int out = 0;
    if (a == kJoin3 - 1) {
        if (f.n_bl > 0) {
            if (b == nb) out = (int)f.n_bl;
            else {
                const int64_t i = (int64_t)b * kBlock;
                const uint64_t key = ((uint64_t)f.contig[i] << 32) | (uint32_t)f.pos[i];
                out = lb_two_level_g<uint64_t>(f.bl, f.bl_coarse, 0, (int)f.n_bl, key);
            }
        }
    } else if (table_present(f, a)) {
        const TrackView& tv = table_view(f, a);
        if (b == nb) out = tv.ptr[f.n_contigs];
        else {
            const int64_t i = (int64_t)b * kBlock;
            const int c = f.contig[i];
            out = lb_two_level_g<int32_t>(tv.starts, tv.coarse, tv.ptr[c], tv.ptr[c + 1], f.pos[i]);
        }
    }
    v.brackets3[gid] = out;
}

// ---- K1 ------------------------------------------------------------------------------------
struct Plan3 {                 // staging plan of one tile (LDS)
    int lo[8], hi[8];          // search brackets per searched array (global indices)
    int base[8];               // global index of the first staged element
    int offS[8], offE[8];      // pool offsets (dwords): starts / ends slices; blacklist in offS[6]
    int cnt[8];                // staged elements
    int bits[8];               // descent depth per array: 2^bits > its search range
    int maxbits;
    int staged;                // every slice fits the pool
};

// PF: the six columns that address the second batch of loads (contig, pos, allele lengths and offsets)
// are fetched one tile ahead, so a tile starts with its window / allele / CSR gathers instead of
// waiting an HBM round trip for their addresses.
template <bool PF>
__global__ __launch_bounds__(kBlock, 4) void featurize3_kernel(const V2Args v) {
    __shared__ uint32_t win[kBlock * kWinStride];                  // 13 KB
    __shared__ int32_t pool[kPool3];                               // 8 KB
    __shared__ __attribute__((aligned(16))) float thr_lds[kThr3];   // 14 KB
    __shared__ uint2 desc_lds[UGVC_N_GROUPS * kMaxFeatures];
    __shared__ int64_t coff_lds[257];
    __shared__ uint8_t css_lds[256];
    __shared__ Plan3 plan;

    const FilterArgs& a = v.f;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int F = UGVC_N_BASE_FEATURES + a.n_tracks;
    const uint8_t* __restrict__ apool = a.alleles;
    const int okbits = (v.pg[0].ok ? 1 : 0) | (v.pg[1].ok ? 2 : 0) | (v.pg[2].ok ? 4 : 0);
    const int gbtbits = (v.pg[0].kind == UGVC_MODEL_GBT ? 1 : 0) | (v.pg[1].kind == UGVC_MODEL_GBT ? 2 : 0) |
                        (v.pg[2].kind == UGVC_MODEL_GBT ? 4 : 0);

    // ---- once per workgroup: model-side tables into LDS
    for (int k = tid; k < UGVC_N_GROUPS * kMaxFeatures; k += kBlock) desc_lds[k] = v.desc3[k];
    // (16-byte pieces: the threshold table is padded to a multiple of four floats on the host)
    for (int k = tid; k < (v.thr_lds_len + 3) / 4; k += kBlock)
        reinterpret_cast<float4*>(thr_lds)[k] = reinterpret_cast<const float4*>(v.thr)[k];
    for (int k = tid; k <= a.n_contigs; k += kBlock) coff_lds[k] = a.contig_off[k];
    css_lds[tid] = v.css_lut[tid];
    __syncthreads();

    // phase clocks (kernel variant bit 6, ugvc_debug_phase_clocks): wave 0 accumulates the core-clock cycles between
    // the phase boundaries of its tiles; slot 6 counts tiles
    __shared__ unsigned long long prof_lds[8];
    const bool prof_on = (a.ablate & 64) != 0 && v.prof != nullptr;
    unsigned long long prof_t = 0;
    if (prof_on && tid == 0) {
        for (int k = 0; k < 8; ++k) prof_lds[k] = 0;
        prof_t = clock64();
    }
    int nc = 0, npos = 0, nrl = 0, nal = 0;
    uint32_t nro = 0, nao = 0;
    auto early = [&](int tile_) {
        const int64_t j_raw = (int64_t)tile_ * kBlock + tid;
        const int64_t j = j_raw < a.n ? j_raw : a.n - 1;
        nc = a.contig[j]; npos = a.pos[j]; nrl = a.ref_len[j]; nal = a.alt_len[j];
        nro = a.ref_off[j]; nao = a.alt_off[j];
    };
    if (PF && (int)blockIdx.x < v.n_blocks) early(blockIdx.x);

    for (int tile = blockIdx.x; tile < v.n_blocks; tile += gridDim.x) {
        const int64_t i_raw = (int64_t)tile * kBlock + tid;
        const bool live = i_raw < a.n;
        const int64_t i = live ? i_raw : a.n - 1;     // idle lanes of the last tile shadow the last variant

        // ---- staging plan (one lane) and variant columns (all lanes), issued together
        if (tid == 0) {
            const int4* r0 = reinterpret_cast<const int4*>(v.brackets3 + (int64_t)tile * 8);
            const int4 l0 = r0[0], l1 = r0[1], h0 = r0[2], h1 = r0[3];
            const int lo[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, 0};
            const int hi[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, 0};
            int used = 0, maxlen = 0, ok = 1;
            // blacklist first (8-byte elements, pool offset stays 8-byte aligned)
            {
                const int t = kJoin3 - 1, na = v.na3[t];
                const int H = hi[t] + 1 < na ? hi[t] + 1 : na;
                plan.lo[t] = lo[t]; plan.hi[t] = hi[t]; plan.base[t] = lo[t];
                plan.cnt[t] = H - lo[t]; plan.offS[t] = used; plan.offE[t] = used;
                used += 2 * (H - lo[t]);
                maxlen = hi[t] - lo[t];
                plan.bits[t] = maxlen > 0 ? 32 - __builtin_clz((unsigned)maxlen) : 0;
            }
            for (int t = 0; t < kJoin3 - 1; ++t) {
                const int na = v.na3[t];
                const int L = lo[t] - 2 > 0 ? lo[t] - 2 : 0;
                const int H = hi[t] + 1 < na ? hi[t] + 1 : na;
                plan.lo[t] = lo[t]; plan.hi[t] = hi[t]; plan.base[t] = L; plan.cnt[t] = H - L;
                plan.offS[t] = used; plan.offE[t] = used + (H - L);
                used += 2 * (H - L);
                const int len = hi[t] - lo[t];
                plan.bits[t] = len > 0 ? 32 - __builtin_clz((unsigned)len) : 0;
                maxlen = len > maxlen ? len : maxlen;
            }
            ok = used <= kPool3;
            plan.staged = ok;
            plan.maxbits = maxlen > 0 ? 32 - __builtin_clz((unsigned)maxlen) : 0;
        }
        int c, pos, rl, al;
        uint32_t ro, ao;
        if (PF) {
            c = nc; pos = npos; rl = nrl; al = nal; ro = nro; ao = nao;
            const int nt = tile + (int)gridDim.x;
            early(nt < v.n_blocks ? nt : tile);
        } else {
            c = a.contig[i]; pos = a.pos[i]; rl = a.ref_len[i]; al = a.alt_len[i];
            ro = a.ref_off[i]; ao = a.alt_off[i];
        }
        const float qual = a.qual[i], sor = a.sor[i];
        const int dp = a.dp[i], adr = a.ad_ref[i], ada = a.ad_alt[i];
        const int gq = a.gq[i];

        // ---- classify_indel (lengths only), then the second batch of loads: reference window,
        // allele bytes, CSR pointers of the side tables
        const bool indel = rl != al;
        const bool ins = rl < al;
        const int classify = !indel ? 0 : (ins ? 1 : 2);
        const int indel_length = ins ? al - rl : rl - al;
        const int64_t clo = coff_lds[c], chi = coff_lds[c + 1];
        const uint32_t clen = (uint32_t)(chi - clo);
        const uint32_t p0 = (uint32_t)(pos - 1);              // 0-based offset inside the contig
        const int64_t g0 = clo + p0;
        int64_t ws = (g0 - 6) & ~(int64_t)15;
        if (ws < 0) ws = 0;
        const int o0 = (int)(g0 - ws);                        // byte of the variant's first base, 6..21 (less at genome start)
        uint32_t* wrow = win + tid * kWinStride;
        {
            // the reference buffer is padded by 64 bytes: three aligned 16-byte loads never overrun
            const uint4* src = reinterpret_cast<const uint4*>(a.ref + ws);
            const uint4 x0 = src[0], x1 = src[1], x2 = src[2];
            wrow[0] = x0.x; wrow[1] = x0.y; wrow[2] = x0.z; wrow[3] = x0.w;
            wrow[4] = x1.x; wrow[5] = x1.y; wrow[6] = x1.z; wrow[7] = x1.w;
            wrow[8] = x2.x; wrow[9] = x2.y; wrow[10] = x2.z; wrow[11] = x2.w;
        }
        // allele bytes: substitutions need ref[0], alt[0]; indels the tail of the longer allele
        const uint32_t lo_off = ins ? ao : ro;
        const int ln = ins ? al : rl;
        uint32_t ab[8];
        {
            const uint32_t q0 = indel ? lo_off + 1 : ro;
            const uint32_t q1 = indel ? lo_off + (2 < ln ? 2 : ln - 1) : ao;
            ab[0] = apool[q0];
            ab[1] = apool[q1];
#pragma unroll
            for (int k = 2; k < 8; ++k) ab[k] = apool[lo_off + (k + 1 < ln ? k + 1 : ln - 1)];
        }
        int plo[kJoin3 - 1], phi[kJoin3 - 1];
#pragma unroll
        for (int t = 0; t < kJoin3 - 1; ++t) {
            plo[t] = phi[t] = 0;
            if (table_present(a, t)) {
                const TrackView& tv = table_view(a, t);
                plo[t] = tv.ptr[(uint32_t)c];
                phi[t] = tv.ptr[(uint32_t)c + 1u];
            }
        }
        __syncthreads();                                      // plan visible
        if (prof_on && tid == 0) { const unsigned long long now_ = clock64(); prof_lds[1] += now_ - prof_t; prof_t = now_; }

        // ---- stage the slices this tile can touch: wave w copies segments w, w+4, w+8, w+12
        const int abl = a.ablate;       // profiling only (ugvc_set_kernel_variant): 2 joins, 4 quantise, 8 window features, 16 append
        const int staged = rfl(plan.staged);
        if (staged && !(abl & 2)) {
            const int wave = tid >> 6;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int seg = wave + 4 * q;                 // 0..5 starts, 6..11 ends, 12 blacklist
                if (seg > 12) break;
                const int t = seg < 6 ? seg : (seg < 12 ? seg - 6 : 6);
                int cnt = rfl(plan.cnt[t]);
                if (cnt <= 0) continue;
                const int32_t* src;
                int dst;
                if (seg == 12) {
                    src = reinterpret_cast<const int32_t*>(a.bl) + 2 * (int64_t)rfl(plan.base[t]);
                    dst = rfl(plan.offS[t]);
                    cnt *= 2;
                } else {
                    const TrackView& tv = table_view(a, t);
                    src = (seg < 6 ? tv.starts : tv.ends) + rfl(plan.base[t]);
                    dst = seg < 6 ? rfl(plan.offS[t]) : rfl(plan.offE[t]);
                }
                const int k0 = lane, k1 = lane + 64;
                const int x0 = k0 < cnt ? src[k0] : 0;
                const int x1 = k1 < cnt ? src[k1] : 0;
                if (k0 < cnt) pool[dst + k0] = x0;
                if (k1 < cnt) pool[dst + k1] = x1;
                for (int k = lane + 128; k < cnt; k += 64) pool[dst + k] = src[k];
            }
        }

        // ---- contig-edge lanes blank the window bytes that lie outside their contig (reads as N)
        if (ws < clo || ws + kWinBytes > chi) {
#pragma unroll
            for (int q = 0; q < kWinDw; ++q) {
                uint32_t m = 0;
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const int64_t gi = ws + 4 * q + bb;
                    m |= (gi >= clo && gi < chi) ? (0xffu << (8 * bb)) : 0u;
                }
                wrow[q] &= m;
            }
        }
        const uint8_t* wb = reinterpret_cast<const uint8_t*>(wrow);
        // reference base at contig offset p0 + d (0 outside the contig); window first, HBM beyond it
        auto ref_at = [&](int d) -> int {
            const int o = o0 + d;
            if (o >= 0 && o < kWinBytes) return wb[o];
            const int64_t gi = g0 + d;
            return (gi >= clo && gi < chi) ? (int)a.ref[gi] : 0;
        };

        // ---- is_hmer_indel.  so = window byte of the first base after the variant's alleles
        // (insertion/substitution: pos+1; deletion: pos+len(ref)); an hmer indel's run starts there.
        const int d_so = (indel && !ins) ? rl : 1;
        const int so = o0 + d_so;
        int hmer_len = 0, hmer_nuc = 0, run = 0;
        if (indel && !(abl & 8)) {
            const int bb = ab[0];
            bool mono = true;
#pragma unroll
            for (int k = 1; k < 8; ++k) mono &= ab[k] == (uint32_t)bb;     // clamped reads repeat the last byte
            if (ln > 9)
                for (int k = 9; k < ln; ++k) mono &= apool[lo_off + k] == bb;
            const uint32_t pstart = p0 + (uint32_t)d_so;
            if (mono && pstart < clen) {
                if (so + 12 <= kWinBytes) {
                    int nrun = 0;
                    bool go = true;
#pragma unroll
                    for (int k = 0; k < 12; ++k) {
                        go = go && wb[so + k] == bb;
                        nrun += go ? 1 : 0;
                    }
                    run = nrun;
                    if (nrun == 12)
                        while (ref_at(d_so + run) == bb && pstart + (uint32_t)run < clen) ++run;
                } else {
                    while (pstart + (uint32_t)run < clen && ref_at(d_so + run) == bb) ++run;
                }
                const uint32_t room = clen - pstart;           // an N run may not run past the contig end
                if ((uint32_t)run > room) run = (int)room;
                if (run > 0) {
                    hmer_len = run + (ins ? 0 : rl - 1);
                    hmer_nuc = bb;
                }
            }
        }
        const bool is_h = indel && hmer_len > 0;
        const int group = !indel ? 0 : (is_h ? 1 : 2);

        // ---- record slots: one returning atomic per wave and group, consumed after the quantisation
        const bool pg_ok = (okbits >> group) & 1;
        const bool mine = live && pg_ok;
        unsigned slot_base = 0, grank = 0;
        {
            const int shard = tile & (kShards - 1);
            unsigned long long m[UGVC_N_GROUPS];
#pragma unroll
            for (int g = 0; g < UGVC_N_GROUPS; ++g) m[g] = __ballot(mine && group == g);
            const unsigned long long mg = lane == 0 ? m[0] : (lane == 1 ? m[1] : m[2]);
            unsigned got = 0;
            if (lane < UGVC_N_GROUPS && mg != 0 && !(abl & 16))
                got = atomicAdd(&v.counters[(lane * kShards + shard) * kCounterStride], (unsigned)__popcll(mg));
            const unsigned long long mm = group == 0 ? m[0] : (group == 1 ? m[1] : m[2]);
            grank = __popcll(mm & ((1ull << lane) - 1));
            slot_base = got;                                  // lanes 0..2 hold the bases; shuffled out at the end
        }

        // ---- get_motif_around (5), gc_content (10)
        int W[11];                                            // bases at pos-5 .. pos+5
#pragma unroll
        for (int k = 0; k < 11; ++k) W[k] = (abl & 8) ? 1 + (k & 3) : wb[o0 - 5 + k];
        if (o0 < 5) {                                         // genome start: the window begins at base 0
#pragma unroll
            for (int k = 0; k < 11; ++k) W[k] = ref_at(k - 5);
        }
        int lmb[kMotif], rmb[kMotif];
        // right motif starts at pos+1 (substitution), pos+len(ref) (non-hmer indel, also a complex
        // insertion with len(ref) > 1) or just past the run (hmer indel: pos+1+hmer_len)
        const int d_r = is_h ? d_so + run : (indel ? rl : 1);
        if (abl & 8) {
#pragma unroll
            for (int k = 0; k < kMotif; ++k) rmb[k] = 2;
        } else if (o0 + d_r + kMotif <= kWinBytes) {
#pragma unroll
            for (int k = 0; k < kMotif; ++k) rmb[k] = wb[o0 + d_r + k];
        } else {
#pragma unroll
            for (int k = 0; k < kMotif; ++k) rmb[k] = ref_at(d_r + k);
        }
        int lm = 0, rm = 0;
        bool motif_n = false;
#pragma unroll
        for (int k = 0; k < kMotif; ++k) {
            lmb[k] = indel ? W[k + 1] : W[k];
            lm = lm * 5 + lmb[k];
            rm = rm * 5 + rmb[k];
            motif_n |= lmb[k] == 0 || rmb[k] == 0;
        }
        int gc_cnt = 0, gc_len = 0;
#pragma unroll
        for (int k = 0; k < kGcWindow; ++k) {
            const uint32_t pw = p0 + 1 - kGcWindow / 2 + k;   // wraps below 0 -> fails the bound test
            const bool inb = pw < clen;
            const int bb = W[k + 1];
            gc_len += inb;
            gc_cnt += inb && bb != 1 && bb != 4;
        }
        const float gc = gc_len > 0 ? (float)((double)gc_cnt / (double)gc_len) : 0.0f;

        // ---- cycle skip
        int css = 3;
        if (!indel && !(abl & 8)) {
            if (rl == 1) {
                const int rb = ab[0], abase = ab[1];
                if (motif_n || rb == 0 || abase == 0) css = 0;
                else css = css_lds[((lmb[kMotif - 1] - 1) << 6) | ((rb - 1) << 4) | ((abase - 1) << 2) | (rmb[0] - 1)];
            } else {
                bool has_n = motif_n;
                for (int k = 0; k < rl; ++k) has_n |= apool[ro + k] == 0 || apool[ao + k] == 0;
                if (has_n) css = 0;
                else {
                    auto mot = [&](const int (&mb)[kMotif], int q) -> int {
                        return mb[0] * (q == 0) + mb[1] * (q == 1) + mb[2] * (q == 2) + mb[3] * (q == 3) + mb[4] * (q == 4);
                    };
                    auto seq_r = [&](int k) -> int {
                        if (k < kMotif) return mot(lmb, k);
                        if (k < kMotif + rl) return apool[ro + k - kMotif];
                        return mot(rmb, k - kMotif - rl);
                    };
                    auto seq_a = [&](int k) -> int {
                        if (k < kMotif) return mot(lmb, k);
                        if (k < kMotif + rl) return apool[ao + k - kMotif];
                        return mot(rmb, k - kMotif - rl);
                    };
                    css = cycle_skip_walk(rl + 2 * kMotif, a.flow, seq_r, seq_a);
                }
            }
        }
        __syncthreads();                                      // staged slices visible
        if (prof_on && tid == 0) { const unsigned long long now_ = clock64(); prof_lds[2] += now_ - prof_t; prof_t = now_; }

        // ---- joins: rank among the starts of every table + blacklist keys, all descents in lock-step
        uint8_t flags = 0;
        bool inside_run = false, close_run = false;
        bool trk[UGVC_MAX_TRACKS] = {false, false, false, false, false};
        const uint64_t key = ((uint64_t)c << 32) | (uint32_t)pos;
        if (abl & 2) {
        } else if (staged) {
            const uint32_t pool_b = lds_addr(pool);                // LDS byte address of the pool
            uint32_t p[kJoin3], pend[kJoin3];
            int baseS[kJoin3];
#pragma unroll
            for (int t = 0; t < kJoin3; ++t) {
                const int blo = rfl(plan.lo[t]), bhi = rfl(plan.hi[t]), bs = rfl(plan.base[t]);
                const int esz = t == kJoin3 - 1 ? 8 : 4;
                int slo = blo, shi = bhi;
                if (t < kJoin3 - 1) {
                    slo = blo > plo[t] ? blo : plo[t];
                    shi = bhi < phi[t] ? bhi : phi[t];
                    if (shi < slo) shi = slo;
                }
                baseS[t] = bs;
                const uint32_t A = pool_b + 4u * (uint32_t)rfl(plan.offS[t]);
                p[t] = A + (uint32_t)(esz * (slo - bs)) - esz;        // address of element slo-1
                pend[t] = A + (uint32_t)(esz * (shi - bs)) - esz;     // address of element shi-1
            }
            // Levels above the shortest non-empty table's depth: only the longer tables take part
            // (wave-uniform branches).  Below it every table steps in lock-step, all reads issued
            // before the first compare so their LDS latencies overlap; empty/absent tables ride
            // along as no-ops (their range test never passes).
            const int bits = rfl(plan.maxbits);
            int tb_[kJoin3], common = 32;
#pragma unroll
            for (int t = 0; t < kJoin3; ++t) {
                tb_[t] = rfl(plan.bits[t]);
                common = tb_[t] > 0 && tb_[t] < common ? tb_[t] : common;
            }
            if (common > bits) common = bits;
            for (int s = bits - 1; s >= common; --s) {
#pragma unroll
                for (int t = 0; t < kJoin3 - 1; ++t) {
                    if (s < tb_[t]) {
                        const uint32_t cand = p[t] + (4u << s);
                        const int x = lds_i32(cand);
                        p[t] = ((int32_t)(pend[t] - cand) >= 0 && x < pos) ? cand : p[t];
                    }
                }
                if (s < tb_[kJoin3 - 1]) {
                    const int t = kJoin3 - 1;
                    const uint32_t cand = p[t] + (8u << s);
                    const uint64_t x = lds_u64(cand);
                    p[t] = ((int32_t)(pend[t] - cand) >= 0 && x < key) ? cand : p[t];
                }
            }
            for (int s = common - 1; s >= 0; --s) {
                uint32_t cand[kJoin3];
                int x[kJoin3 - 1];
#pragma unroll
                for (int t = 0; t < kJoin3 - 1; ++t) {
                    cand[t] = p[t] + (4u << s);
                    x[t] = lds_i32(cand[t]);
                }
                cand[kJoin3 - 1] = p[kJoin3 - 1] + (8u << s);
                const uint64_t xk = lds_u64(cand[kJoin3 - 1]);
#pragma unroll
                for (int t = 0; t < kJoin3 - 1; ++t)
                    p[t] = ((int32_t)(pend[t] - cand[t]) >= 0 && x[t] < pos) ? cand[t] : p[t];
                p[kJoin3 - 1] = ((int32_t)(pend[kJoin3 - 1] - cand[kJoin3 - 1]) >= 0 && xk < key) ? cand[kJoin3 - 1] : p[kJoin3 - 1];
            }
#pragma unroll
            for (int t = 0; t < kJoin3 - 1; ++t) {
                if (!table_present(a, t)) continue;
                const uint32_t A = pool_b + 4u * (uint32_t)rfl(plan.offS[t]);
                const uint32_t dE = 4u * (uint32_t)(rfl(plan.offE[t]) - rfl(plan.offS[t]));
                const int sg = baseS[t] + (int)((p[t] + 4 - A) >> 2);     // #starts < pos (global index)
                const int e1v = lds_i32(p[t] + dE);                       // ends[sg-1]
                const int e2v = lds_i32(p[t] + dE - 4);                   // ends[sg-2]
                const bool valid = sg > plo[t];
                if (t == 0) {
                    // runs are disjoint: #ends < pos is sg-1 or sg
                    const bool ins_run = valid && e1v >= pos;
                    if (phi[t] > plo[t]) {
                        const int eg = valid ? sg - 1 + (e1v < pos ? 1 : 0) : sg;
                        const int D = a.hpol_dist;
                        auto near = [&](int x) { const int d = pos - x; return (d < 0 ? -d : d) < D; };
                        const int s1v = lds_i32(p[t]);                    // starts[sg-1]
                        const int s0v = lds_i32(p[t] + 4);                // starts[sg]
                        bool cd = (valid && near(s1v)) || (sg <= phi[t] - 1 && near(s0v));
                        const uint32_t pe = p[t] + dE + 4u * (uint32_t)(eg - (sg - 1));   // address of ends[eg]
                        const int ee = lds_i32(pe), em = lds_i32(pe - 4);
                        cd = cd || (eg - 1 >= plo[t] && near(em)) || (eg <= phi[t] - 1 && near(ee));
                        inside_run = ins_run;
                        close_run = cd && !ins_run;
                    }
                } else {
                    const bool in = valid && e1v >= pos && (sg - 1 == plo[t] || e2v < pos);
                    trk[t - 1] = in;
                    flags |= in ? (uint8_t)(1u << (UGVC_FLAG_TRACK0_SHIFT + t - 1)) : 0;
                }
            }
            if (a.n_bl > 0) {
                const int t = kJoin3 - 1;
                const uint32_t A = pool_b + 4u * (uint32_t)rfl(plan.offS[t]);
                const int r = (int)((p[t] + 8 - A) >> 3);                 // local rank
                const uint64_t x = lds_u64(p[t] + 8);
                if (r < rfl(plan.cnt[t]) && x == key) flags |= UGVC_FLAG_COHORT_FP;
            }
        } else {
            // dense tile (slices exceed the pool): same arithmetic on the HBM copies
#pragma unroll
            for (int t = 0; t < kJoin3 - 1; ++t) {
                if (!table_present(a, t)) continue;
                const TrackView& tv = table_view(a, t);
                const int sg = lb_i32_g(tv.starts, plo[t], phi[t], pos);
                const bool valid = sg > plo[t];
                const int e1v = valid ? tv.ends[sg - 1] : 0;
                if (t == 0) {
                    const bool ins_run = valid && e1v >= pos;
                    if (phi[t] > plo[t]) {
                        const int eg = valid ? sg - 1 + (e1v < pos ? 1 : 0) : sg;
                        const int D = a.hpol_dist;
                        auto near = [&](int x) { const int d = pos - x; return (d < 0 ? -d : d) < D; };
                        bool cd = (valid && near(tv.starts[sg - 1])) || (sg <= phi[t] - 1 && near(tv.starts[sg]));
                        cd = cd || (eg - 1 >= plo[t] && near(tv.ends[eg - 1])) || (eg <= phi[t] - 1 && near(tv.ends[eg]));
                        inside_run = ins_run;
                        close_run = cd && !ins_run;
                    }
                } else {
                    const bool in = valid && e1v >= pos && (sg - 1 == plo[t] || tv.ends[sg - 2] < pos);
                    trk[t - 1] = in;
                    flags |= in ? (uint8_t)(1u << (UGVC_FLAG_TRACK0_SHIFT + t - 1)) : 0;
                }
            }
            if (a.n_bl > 0) {
                const int r = lb_u64_g(a.bl, 0, (int)a.n_bl, key);
                if (r < (int)a.n_bl && a.bl[r] == key) flags |= UGVC_FLAG_COHORT_FP;
            }
        }
        if (a.mark_hpol && (inside_run || close_run)) flags |= UGVC_FLAG_HPOL_RUN;
        if (live) a.flags[i] = flags;
        if (!pg_ok && live) {              // no model for this variant type: score 0, PASS
            a.score[i] = 0.f;
            a.filter[i] = UGVC_FILTER_PASS;
        }

        if (prof_on && tid == 0) { const unsigned long long now_ = clock64(); prof_lds[3] += now_ - prof_t; prof_t = now_; }
        // ---- quantise: feature -> rank among the group's sorted thresholds
        const float vaf = dp > 0 ? __fdiv_rn((float)ada, (float)dp) : 0.0f;
        const uint2* dsc = desc_lds + group * kMaxFeatures;
        uint32_t c0 = 0, c1 = 0, c2 = 0;
        // the dword of a feature is the same for every group (host: joint_layout), so the
        // accumulator is chosen with wave-uniform masks; only the bit offset is per lane
        auto put = [&](int j, uint32_t dy, uint32_t code) {
            const uint32_t val = code << ((dy >> 18) & 31);
            const uint32_t dwj = v.dw3[j];
            c0 |= val & (dwj == 0 ? ~0u : 0u);
            c1 |= val & (dwj == 1 ? ~0u : 0u);
            c2 |= val & (dwj == 2 ? ~0u : 0u);
        };
        const bool upper = (gbtbits >> group) & 1;
        if (pg_ok && !(abl & 4)) {
            // float features: lock-step descent over the LDS threshold slices
            {
                const float fx[4] = {qual, sor, vaf, gc};
                const int fj[4] = {0, 1, 5, 13};
                const uint32_t thr_b = lds_addr(thr_lds);
                uint32_t q[4], qend[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint2 d = dsc[fj[k]];
                    q[k] = thr_b + 4u * (d.x & 0xFFFFFu) - 4;
                    qend[k] = q[k] + 4u * (d.y & 0xFFFFu);
                }
                const int fb0 = v.thr_bits4[0], fb1 = v.thr_bits4[1], fb2 = v.thr_bits4[2], fb3 = v.thr_bits4[3];
                const int fbm = max(max(fb0, fb1), max(fb2, fb3));
                // an empty slice (feature unused by the lane's group) never passes the range test
                for (int s = fbm - 1; s >= 0; --s) {
                    uint32_t cand[4];
                    float t[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        cand[k] = q[k] + (4u << s);
                        t[k] = lds_f32(cand[k]);
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const bool lt = upper ? t[k] <= fx[k] : t[k] < fx[k];
                        q[k] = ((int32_t)(qend[k] - cand[k]) >= 0 && lt) ? cand[k] : q[k];
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint2 d = dsc[fj[k]];
                    uint32_t cd = (q[k] + 4 - (thr_b + 4u * (d.x & 0xFFFFFu))) >> 2;
                    if (fx[k] != fx[k]) cd = d.y & 0xFFFFu;       // NaN compares false: always the right branch
                    put(fj[k], d.y, cd);
                }
            }
            // integer-valued features: one LUT load each, issued in two batches of independent loads
            int iv[kMaxFeatures];
            iv[0] = iv[1] = iv[5] = iv[13] = 0;
            iv[2] = dp; iv[3] = adr; iv[4] = ada; iv[6] = gq; iv[7] = classify; iv[8] = indel_length;
            iv[9] = hmer_len; iv[10] = hmer_nuc; iv[11] = lm; iv[12] = rm; iv[14] = css;
#pragma unroll
            for (int j = 15; j < kMaxFeatures; ++j) iv[j] = 0;
            {
                // the seven 0/1 features sit in fixed bits 25..31 of dword 2; rank code == value
                // wherever the group's model tests them below 1 (boolmask3), else constant 0
                const uint32_t bits7 = (inside_run ? 1u : 0u) | (close_run ? 2u : 0u) | ((uint32_t)(flags >> UGVC_FLAG_TRACK0_SHIFT) << 2);
                const uint32_t bm = group == 0 ? v.boolmask3[0] : (group == 1 ? v.boolmask3[1] : v.boolmask3[2]);
                c2 |= (bits7 & bm) << 25;
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int j0 = half == 0 ? 2 : 9, j1 = half == 0 ? 9 : 15;
                uint32_t code[kMaxFeatures];
                bool slow = false;
#pragma unroll
                for (int j = j0; j < j1; ++j) {
                    if (j == 5 || j == 13) continue;
                    if (j >= F) break;
                    const uint2 d = dsc[j];
                    const uint32_t len = d.y & 0xFFFFu;
                    const uint32_t x = (uint32_t)iv[j];
                    const uint32_t idx = x < len ? x : len - 1;
                    code[j] = v.lut[(d.x & 0xFFFFFu) + idx];
                    slow |= x >= len && ((d.x >> 30) == 1 || (int32_t)x < 0);        // past a cut table, or negative
                }
                if (slow) {                                       // value beyond the LUT: search the thresholds in HBM
#pragma unroll
                    for (int j = j0; j < j1; ++j) {
                        if (j == 5 || j == 13) continue;
                        if (j >= F) break;
                        const uint2 d = dsc[j];
                        if ((uint32_t)iv[j] >= (d.y & 0xFFFFu) && ((d.x >> 30) == 1 || iv[j] < 0)) {
                            const FeatDesc fd = v.desc[group * kMaxFeatures + j];
                            const uint32_t toff = fd.thr & 0xFFFFF, tlen = fd.thr >> 20;
                            const float x = (float)iv[j];
                            uint32_t bb = 0, len = tlen;
                            while (len > 0) {
                                const uint32_t hf = len >> 1;
                                const float t = v.thr[toff + bb + hf];
                                const bool lt = upper ? t <= x : t < x;
                                bb = lt ? bb + hf + 1 : bb;
                                len = lt ? len - hf - 1 : hf;
                            }
                            code[j] = bb;
                        }
                    }
                }
#pragma unroll
                for (int j = j0; j < j1; ++j) {
                    if (j == 5 || j == 13) continue;
                    if (j >= F) break;
                    put(j, dsc[j].y, code[j]);
                }
            }
        }

        if (prof_on && tid == 0) { const unsigned long long now_ = clock64(); prof_lds[4] += now_ - prof_t; prof_t = now_; }
        // ---- append {codes, variant index} to the group's sharded record list
        {
            const unsigned b0 = __shfl(slot_base, 0), b1 = __shfl(slot_base, 1), b2 = __shfl(slot_base, 2);
            const unsigned sb = group == 0 ? b0 : (group == 1 ? b1 : b2);
            const int shard = tile & (kShards - 1);
            if (mine && !(abl & 16))
                v.records[group][(size_t)shard * v.shard_cap + sb + grank] = make_uint4(c0, c1, c2, (uint32_t)i);
        }
        __syncthreads();                                      // plan / pool are rewritten by the next tile
        if (prof_on && tid == 0) { const unsigned long long now_ = clock64(); prof_lds[5] += now_ - prof_t; prof_t = now_; prof_lds[6] += 1; }
    }
    if (prof_on && tid == 0) {
        for (int k = 0; k < 7; ++k) atomicAdd(&v.prof[k], prof_lds[k]);
    }
}

// ---- K2 ------------------------------------------------------------------------------------
// One workgroup per CU, one variant-type group per workgroup, whole forest in LDS (layout as v2:
// u32 nodes in 1-based heap order, rank[0:16) | code-plane byte offset[16:32)).

template <bool FAST, int NTM>
__global__ __launch_bounds__(kK2Threads) void forest3_kernel(const V2Args v) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned shard_off[kShards + 1];
    __shared__ unsigned totals[UGVC_N_GROUPS];
    __shared__ uint32_t pdesc[64];                               // the group's plane descriptors (F <= 32 features)
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int n_waves = blockDim.x >> 6;
    if (tid < UGVC_N_GROUPS) totals[tid] = 0;
    __syncthreads();
    for (int k = tid; k < UGVC_N_GROUPS * kShards; k += blockDim.x) {
        const unsigned cshard = v.counters[k * kCounterStride];
        if (cshard) atomicAdd(&totals[k / kShards], cshard);
    }
    __syncthreads();
    // workgroups are split over the groups in proportion to count x trees x depth
    const int B = gridDim.x;
    unsigned cnt[UGVC_N_GROUPS];
    double work[UGVC_N_GROUPS], tot = 0.0;
    for (int g = 0; g < UGVC_N_GROUPS; ++g) {
        cnt[g] = v.pg[g].ok ? totals[g] : 0u;
        work[g] = (double)cnt[g] * v.pg[g].T * v.pg[g].D;
        tot += work[g];
    }
    if (tot == 0.0) return;
    int nb[UGVC_N_GROUPS], used = 0, big = 0;
    for (int g = 0; g < UGVC_N_GROUPS; ++g) {
        nb[g] = cnt[g] ? (int)(B * (work[g] / tot) + 0.5) : 0;
        if (cnt[g] && nb[g] < 1) nb[g] = 1;
        used += nb[g];
        if (work[g] > work[big]) big = g;
    }
    nb[big] += B - used;
    if (nb[big] < 1) return;
    int g = 0, lb = blockIdx.x;
    while (g < UGVC_N_GROUPS - 1 && lb >= nb[g]) { lb -= nb[g]; ++g; }
    // the split came through LDS, so the compiler holds g and everything read through it (D, T, table
    // pointers, the per-tree bases of the walk) in VGPRs and spends ~20 % more vector instructions on
    // wave-uniform arithmetic: pin them to SGPRs here
    g = rfl(g);
    lb = rfl(lb);
    const int nbg = rfl(nb[g]);
    const PackedGroupView pg = v.pg[g];
    const unsigned n = (unsigned)rfl((int)cnt[g]);
    if (n == 0 || nbg == 0) return;

    // the descriptors go to LDS once: read from global memory per chunk they sit, as vector loads, in front of every
    // chunk's decode and make it wait for the record prefetched for the NEXT chunk as well (in-order vmcnt)
    if (tid < 64) pdesc[tid] = tid < pg.n_planes ? pg.plane_desc[tid] : 0u;
    if (wave == 0) {                                             // exclusive scan of the group's shard counts
        unsigned x[4], s = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { x[k] = v.counters[(g * kShards + lane * 4 + k) * kCounterStride]; s += x[k]; }
        unsigned incl = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned y = __shfl_up(incl, d);
            if (lane >= d) incl += y;
        }
        unsigned runv = incl - s;
#pragma unroll
        for (int k = 0; k < 4; ++k) { shard_off[lane * 4 + k] = runv; runv += x[k]; }
        if (lane == 63) shard_off[kShards] = runv;
    }

    const int D = pg.D, NL = 1 << D;
    const size_t n_nodes = (size_t)pg.T * NL;
    uint32_t* nodes = reinterpret_cast<uint32_t*>(smem);
    size_t off = 0;
    double2* pairs = nullptr;
    float* leaf_f32 = nullptr;
    uint16_t* leaf_idx = nullptr;
    uint2* last4 = nullptr;
    double* p1 = nullptr;
    const int H = NL >> 1;
    if (FAST) {
        const size_t n_hi = (size_t)pg.T * H;
        // the three tables are padded to 16 bytes (host and LDS): the fill is a handful of independent 16-byte
        // loads per thread instead of a dependent load per dword (it is pure latency in front of every launch)
        const size_t b_hi = (n_hi * 4 + 15) & ~(size_t)15, b_last = (n_hi * 8 + 15) & ~(size_t)15;
        const size_t b_p1 = ((size_t)pg.n_pairs * 8 + 15) & ~(size_t)15;
        off = b_hi;
        last4 = reinterpret_cast<uint2*>(smem + off);
        off += b_last;
        p1 = reinterpret_cast<double*>(smem + off);
        off += b_p1;
        {
            const uint4* s0 = reinterpret_cast<const uint4*>(pg.hi4);
            const uint4* s1 = reinterpret_cast<const uint4*>(pg.last4);
            const uint4* s2 = reinterpret_cast<const uint4*>(pg.p1);
            uint4* d0 = reinterpret_cast<uint4*>(smem);
            uint4* d1 = reinterpret_cast<uint4*>(smem + b_hi);
            uint4* d2 = reinterpret_cast<uint4*>(smem + b_hi + b_last);
            const size_t n0 = b_hi / 16, n1 = b_last / 16, n2 = b_p1 / 16;
            for (size_t k = tid; k < n0 + n1 + n2; k += blockDim.x) {
                if (k < n0) d0[k] = s0[k];
                else if (k < n0 + n1) d1[k - n0] = s1[k - n0];
                else d2[k - n0 - n1] = s2[k - n0 - n1];
            }
        }
    } else {
        off = (n_nodes * 4 + 15) & ~(size_t)15;
        pairs = reinterpret_cast<double2*>(smem + off);
        leaf_f32 = reinterpret_cast<float*>(smem + off);
        off += pg.kind == UGVC_MODEL_RF ? (size_t)pg.n_pairs * 16 : ((n_nodes * 4 + 15) & ~(size_t)15);
        leaf_idx = reinterpret_cast<uint16_t*>(smem + off);
        if (pg.kind == UGVC_MODEL_RF) off += (n_nodes * 2 + 15) & ~(size_t)15;
        for (size_t k = tid; k < n_nodes; k += blockDim.x) nodes[k] = pg.nodes[k];
        if (pg.kind == UGVC_MODEL_RF) {
            for (size_t k = tid; k < (size_t)pg.n_pairs; k += blockDim.x) pairs[k] = pg.pairs[k];
            const uint32_t* src = reinterpret_cast<const uint32_t*>(pg.leaf_idx);      // T * 2^D halfwords: an even count
            uint32_t* dst = reinterpret_cast<uint32_t*>(leaf_idx);
            for (size_t k = tid; k < n_nodes / 2; k += blockDim.x) dst[k] = src[k];
        } else {
            for (size_t k = tid; k < n_nodes; k += blockDim.x) leaf_f32[k] = pg.leaf_f32[k];
        }
    }
    uint16_t* planes_all = reinterpret_cast<uint16_t*>(smem + off);
    __syncthreads();

    const int P = pg.n_planes;
    uint16_t* planes = planes_all + (size_t)wave * P * 64;
    // lane -> halfword slot of a plane: lanes 0..31 take the low halves of the 32 dwords, lanes
    // 32..63 the high halves, so both lane groups of a ds_read_u16 hit 32 distinct banks
    const int hslot = ((lane & 31) << 1) | (lane >> 5);
    const uint32_t nodes_b = lds_addr(nodes);
    const uint32_t planes_lane_b = lds_addr(planes + hslot);
    const unsigned waves = (unsigned)nbg * n_waves;
    const uint4* __restrict__ rec = v.records[g];
    const int T = pg.T;
    // chunk slots are wave-major over the group's workgroups: the last, partial round then leaves a
    // few waves busy on every CU instead of all sixteen on a few CUs
    // the record of a chunk is fetched one chunk ahead: the shard search and the HBM round trip of the
    // next 64 records run under the current chunk's walk
    auto fetch = [&](unsigned chunk, bool& live) -> uint4 {
        const unsigned r = chunk * 64 + lane;
        live = r < n;
        const unsigned rr = live ? r : n - 1;
        int lo = 0, len = kShards;
        while (len > 1) {                                        // shard of record rr
            const int half = len >> 1;
            const bool ge = shard_off[lo + half] <= rr;
            lo = ge ? lo + half : lo;
            len = ge ? len - half : half;
        }
        return rec[(size_t)lo * v.shard_cap + (rr - shard_off[lo])];
    };
    unsigned chunk = (unsigned)rfl(wave) * (unsigned)nbg + (unsigned)lb;
    bool live_next = false;
    uint4 q_next = make_uint4(0, 0, 0, 0);
    if ((uint64_t)chunk * 64 < n) q_next = fetch(chunk, live_next);
    for (; (uint64_t)chunk * 64 < n; chunk += waves) {
        const uint4 q = q_next;
        const bool live = live_next;
        if ((uint64_t)(chunk + waves) * 64 < n) q_next = fetch(chunk + waves, live_next);
        for (int p = 0; p < P; ++p) {
            const uint32_t pd = (uint32_t)rfl((int)pdesc[p]);    // dword[0:2) | bit_off[2:7) | width[7:11)
            const uint32_t dw = pd & 3;
            const uint32_t word = dw == 0 ? q.x : (dw == 1 ? q.y : q.z);
            planes[p * 64 + hslot] = (uint16_t)__builtin_amdgcn_ubfe(word, (pd >> 2) & 31, (pd >> 7) & 15);
        }
        double a0 = 0.0, a1 = 0.0;
        float margin = pg.base;
        float score;
        uint8_t filt;
        if (FAST) {
            const uint32_t hi_b = nodes_b, last_b = lds_addr(last4), p1_b = lds_addr(p1);
            int t = 0;
            if (NTM > 8) {
                for (; t + NTM <= T; t += NTM) {
                    uint32_t pi[NTM];
                    walk4<NTM>(hi_b, last_b, planes_lane_b, t, D, H, pi);
                    double pv[NTM];
#pragma unroll
                    for (int k = 0; k < NTM; ++k) pv[k] = lds_f64(p1_b + 8u * pi[k]);
#pragma unroll
                    for (int k = 0; k < NTM; ++k) a1 += pv[k];
                }
            }
            for (; t + 8 <= T; t += 8) {
                uint32_t pi[8];
                walk4<8>(hi_b, last_b, planes_lane_b, t, D, H, pi);
                double pv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) pv[k] = lds_f64(p1_b + 8u * pi[k]);
#pragma unroll
                for (int k = 0; k < 8; ++k) a1 += pv[k];
            }
            for (; t < T; ++t) {
                uint32_t pi[1];
                walk4<1>(hi_b, last_b, planes_lane_b, t, D, H, pi);
                a1 += lds_f64(p1_b + 8u * pi[0]);
            }
            const double half = 0.5 * (double)T, band = pg.band;
            const double pr1 = a1 / (double)T;
            score = (float)pr1;
            filt = a1 > half ? UGVC_FILTER_PASS : UGVC_FILTER_LOW_SCORE;
            // inside the band around T/2 (rounding of the two sums, see model_pack.hip) the class-0 sum
            // decides as scikit-learn's argmax does: redo the walk with both payload sums, in tree order
            // (wave-uniform, exact ties only in practice)
            if (__builtin_amdgcn_ballot_w64(fabs(a1 - half) <= band) != 0) {
                double b0 = 0.0, b1 = 0.0;
                int tt = 0;
                for (; tt + 8 <= T; tt += 8) {
                    uint32_t pi[8];
                    walk4<8>(hi_b, last_b, planes_lane_b, tt, D, H, pi);
                    double2 pv[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) pv[k] = pg.pairs[pi[k]];
#pragma unroll
                    for (int k = 0; k < 8; ++k) { b0 += pv[k].x; b1 += pv[k].y; }
                }
                for (; tt < T; ++tt) {
                    uint32_t pi[1];
                    walk4<1>(hi_b, last_b, planes_lane_b, tt, D, H, pi);
                    const double2 pv = pg.pairs[pi[0]];
                    b0 += pv.x; b1 += pv.y;
                }
                const double q0 = b0 / (double)T, q1 = b1 / (double)T;
                if (fabs(a1 - half) <= band) filt = q1 > q0 ? UGVC_FILTER_PASS : UGVC_FILTER_LOW_SCORE;
            }
        } else {
        int t = 0;
        for (; t + 8 <= T; t += 8) {
            int leaf[8];
            walk3<8>(nodes_b, planes_lane_b, t, D, NL, leaf);
            if (pg.kind == UGVC_MODEL_RF) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { const double2 pv = pairs[leaf_idx[leaf[k]]]; a0 += pv.x; a1 += pv.y; }
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) margin += leaf_f32[leaf[k]];
            }
        }
        for (; t < T; ++t) {
            int leaf[1];
            walk3<1>(nodes_b, planes_lane_b, t, D, NL, leaf);
            if (pg.kind == UGVC_MODEL_RF) { const double2 pv = pairs[leaf_idx[leaf[0]]]; a0 += pv.x; a1 += pv.y; }
            else margin += leaf_f32[leaf[0]];
        }
        if (pg.kind == UGVC_MODEL_RF) {
            const double pr0 = a0 / (double)T, pr1 = a1 / (double)T;
            score = (float)pr1;
            filt = pr1 > pr0 ? UGVC_FILTER_PASS : UGVC_FILTER_LOW_SCORE;
        } else {
            score = 1.0f / (1.0f + expf(-margin));
            filt = margin > 0.0f ? UGVC_FILTER_PASS : UGVC_FILTER_LOW_SCORE;
        }
        }
        if (live) {
            v.f.score[q.w] = score;
            v.f.filter[q.w] = filt;
        }
    }
}

static size_t k3_lds_bytes(const PackedGroupView& pg, int n_waves, bool fast) {
    const size_t NL = (size_t)1 << pg.D, n_nodes = (size_t)pg.T * NL;
    if (fast) {
        const size_t n_hi = n_nodes / 2;
        return ((n_hi * 4 + 15) & ~(size_t)15) + ((n_hi * 8 + 15) & ~(size_t)15) + (((size_t)pg.n_pairs * 8 + 15) & ~(size_t)15) +
               (size_t)n_waves * pg.n_planes * 128;
    }
    size_t b = (n_nodes * 4 + 15) & ~(size_t)15;
    if (pg.kind == UGVC_MODEL_RF) b += (size_t)pg.n_pairs * 16 + ((n_nodes * 2 + 15) & ~(size_t)15);
    else b += (n_nodes * 4 + 15) & ~(size_t)15;
    return b + (size_t)n_waves * pg.n_planes * 128;
}

int launch_filter_v3(ugvc_ctx* ctx, const FilterArgs& a) {
    if (a.n == 0) return 0;
    V2Args v;
    v.f = a;
    if (v2_fill_args(ctx, v, a.n)) return -1;
    const int64_t nbr = std::max<int64_t>((int64_t)(v.n_blocks + 1) * 8, UGVC_N_GROUPS * kShards);
    UGVC_LAUNCH(bracket3_kernel, dim3((unsigned)((nbr + 255) / 256)), dim3(256), 0, ctx->stream, v);
    // profiling knobs (ugvc_set_kernel_variant): bits 12-13 of the kernel variant cap K1's workgroups per CU,
    // bits 14-15 pick K2's wave count; 0 = the defaults
    const int k1_bpc = ((a.ablate >> 12) & 3) ? ((a.ablate >> 12) & 3) : 4;
    const int k1_grid = std::min(v.n_blocks, ctx->n_cus * k1_bpc);
    // kernel variant bit 5 (32): K1 without the one-tile-ahead column prefetch
    if (a.ablate & 32) UGVC_LAUNCH(featurize3_kernel<false>, dim3((unsigned)k1_grid), dim3(kBlock), 0, ctx->stream, v);
    else UGVC_LAUNCH(featurize3_kernel<true>, dim3((unsigned)k1_grid), dim3(kBlock), 0, ctx->stream, v);
    return launch_forest3(ctx, v, a);
}

// K2 over the record lists K1 (v3 or v4) appended
int launch_forest3(ugvc_ctx* ctx, const V2Args& v, const FilterArgs& a) {
    const int k2_pick = (a.ablate >> 14) & 3;
    int n_waves = 0;
    size_t lds = 0;
    // single-sum RF kernel when every uploaded group allows it (kernel variant bit 10 forces the pair kernel)
    bool fast = !(a.ablate & 1024);
    for (int g = 0; g < UGVC_N_GROUPS; ++g)
        if (v.pg[g].ok && !v.pg[g].fast4) fast = false;
    for (int w : {16, 12, 8, 4}) {
        if (k2_pick && w > (k2_pick == 1 ? 12 : (k2_pick == 2 ? 8 : 4))) continue;
        size_t need = 0;
        for (int g = 0; g < UGVC_N_GROUPS; ++g)
            if (v.pg[g].ok) need = std::max(need, k3_lds_bytes(v.pg[g], w, fast));
        if (need + 2048 <= 160 * 1024) { n_waves = w; lds = need; break; }
    }
    if (n_waves == 0) return fail("internal: packed forest does not fit LDS");
    if (lds && !(a.ablate & 1)) {
        using K2 = void (*)(const V2Args);
        // trees in flight per lane: 8 (16 measured slower: 345 vs 324 us per 5M pass - the LDS pipeline, not
        // the dependent latency, bounds the walk); kernel variant bit 11 selects 16
        const K2 fn = !fast ? forest3_kernel<false, 8> : ((a.ablate & 2048) ? forest3_kernel<true, 16> : forest3_kernel<true, 8>);
        static bool attr_set[64] = {};                           // (function attributes are per device, as in kernels_v5.hip)
        if (!attr_set[ctx->device & 63]) {
            for (K2 f : {(K2)forest3_kernel<false, 8>, (K2)forest3_kernel<true, 8>, (K2)forest3_kernel<true, 16>})
                UGVC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
            attr_set[ctx->device & 63] = true;
        }
        UGVC_LAUNCH(fn, dim3((unsigned)ctx->n_cus), dim3(n_waves * 64), lds, ctx->stream, v);
    }
    UGVC_HIP(hipGetLastError());
    return 0;
}

}  // namespace ugvc
