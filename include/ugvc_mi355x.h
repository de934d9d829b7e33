/* ugvc_mi355x.h - C ABI of the MI355X-native post-GATK variant-filtering engine.
 *
 * Drop-in boundary for the hot path of `ugvc filter_variants_pipeline` /
 * `train_models_pipeline` (featurize -> interval/blacklist lookup -> tree-ensemble score ->
 * FILTER).  The reference has NO native boundary for this path: its contract is a Python
 * module with run(argv) (/root/reference/ugvc/__main__.py:42-56,103-105) whose body calls, per
 * VCF, the pandas functions listed next to each entry point below (bodies live in the
 * un-vendored submodule ugbio_utils; cited by call site).  This header is therefore the
 * BUILDER-DEFINED FFI a maintainer binds with ctypes: plain pointers and
 * sizes, caller owns every host buffer, opaque context handle, int return codes.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; ugvc_last_error() gives the message
 *     (thread-local).  Reference behaviour is Python exceptions (e.g.
 *     ugvc/comparison/quick_fingerprinter.py:134-135); the Python shim raises RuntimeError.
 *   - all table pointers are HOST pointers to contiguous column arrays (numpy); the library
 *     copies to HBM and never frees or retains caller memory.
 *   - one host thread per context; kernels run on the context's own HIP stream; calls are
 *     synchronous unless stated.
 *   - base codes: N=0 A=1 C=2 G=3 T=4.  POS is the 1-based VCF position.  Variants must be
 *     sorted by (contig, pos).
 */
#ifndef UGVC_MI355X_H
#define UGVC_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UGVC_ABI_VERSION 2          /* 2: contig column u16 (a WGS reference has 3 366 contigs:
                                     test/resources/unit/vcfbed/test_vcftools/header.txt) */
#define UGVC_MAX_TRACKS 5
#define UGVC_N_GROUPS 3          /* snp, h-indel, non-h-indel */
#define UGVC_N_BASE_FEATURES 17  /* + one per annotation track; order: schema.BASE_FEATURES */
#define UGVC_MODEL_RF 0          /* sklearn forest: x <= thr, mean of f64 leaf fractions      */
#define UGVC_MODEL_GBT 1         /* XGBoost-style:  x <  thr, f32 additive margin + sigmoid   */

#define UGVC_FILTER_PASS 0
#define UGVC_FILTER_LOW_SCORE 1
#define UGVC_FLAG_HPOL_RUN 1
#define UGVC_FLAG_COHORT_FP 2
#define UGVC_FLAG_SEC 4
#define UGVC_FLAG_TRACK0_SHIFT 3

typedef struct ugvc_ctx ugvc_ctx;

/* Variant table, one column per array (replaces the DataFrame built by
 * ugbio_core.vcfbed.vcftools.get_vcf_df: call sites ugvc/pipelines/run_no_gt_report.py:307-312,
 * shape ugvc/reports/report_wo_gt.ipynb:1207-1210). */
typedef struct ugvc_variants {
    int64_t n;
    const uint16_t* contig;   /* contig index into the uploaded reference (<= 65535 contigs) */
    const int32_t* pos;       /* 1-based                                                   */
    const uint16_t* ref_len;
    const uint16_t* alt_len;
    const uint32_t* ref_off;  /* offsets into `alleles`                                    */
    const uint32_t* alt_off;
    const uint8_t* alleles;   /* pool of base codes                                        */
    int64_t alleles_len;
    const float* qual;        /* QUAL (or 10*TLOD for --is_mutect, set by the host)        */
    const float* sor;         /* INFO/SOR                                                  */
    const int32_t* dp;        /* FORMAT/DP                                                 */
    const int32_t* ad_ref;    /* FORMAT/AD[0]                                              */
    const int32_t* ad_alt;    /* FORMAT/AD[1]                                              */
    const uint8_t* gq;        /* FORMAT/GQ                                                 */
} ugvc_variants;

/* Per-variant outputs (replaces the FILTER / TREE_SCORE / HPOL_RUN / COHORT_FP columns the
 * reference writes back per record: docs/howto-callset-filter.md:65,
 * ugvc/pipelines/evaluate_concordance.py:47). */
typedef struct ugvc_results {
    float* tree_score;        /* TREE_SCORE                                                */
    uint8_t* filter;          /* UGVC_FILTER_*                                             */
    uint8_t* flags;           /* UGVC_FLAG_* | track bits                                  */
} ugvc_results;

/* Pileup tally outputs (builder-defined; the reference reads FORMAT/AD, DP, SB, VAF, INFO/SOR
 * pre-computed by the caller: test/resources/unit/vcfbed/test_vcftools/header.txt:3379,3391-3398). */
typedef struct ugvc_pileup_out {
    int32_t* ref_fwd; int32_t* ref_rev; int32_t* alt_fwd; int32_t* alt_rev;
    int32_t* other;   int32_t* dp;      int32_t* bq_ref;  int32_t* bq_alt;
    float* vaf;       float* sor;
} ugvc_pileup_out;

int ugvc_abi_version(void);
const char* ugvc_last_error(void);

/* ---- context ------------------------------------------------------------------------- */
int ugvc_ctx_create(int device_id, ugvc_ctx** out);
int ugvc_ctx_destroy(ugvc_ctx* ctx);
int ugvc_device_info(ugvc_ctx* ctx, char* name, int name_cap, int* n_cus, int64_t* hbm_bytes);
/* One integer property of the context's device: what = 0 peak engine clock in kHz (hipDeviceProp_t::clockRate), 1 compute
 * units, 2 peak memory clock in kHz, 3 LDS bytes per workgroup.  bench.py prices its instruction-issue floor with it. */
int ugvc_device_attr(ugvc_ctx* ctx, int what, int64_t* out);
int ugvc_sync(ugvc_ctx* ctx);
/* Device canary (builder-defined; the reference's cheapest self-check is the CI's `-h` smoke of the two pipelines,
 * .github/workflows/python-package-conda.yml:46-61): n 64-bit words host -> device -> host, then the library's prefix-sum
 * kernel over them against the host's sum.  0 = this device copies and computes; the error text names the step that
 * did not.  Debug allocation modes of every device buffer (tests only, results unchanged): UGVC_GUARD=1|2 (end | start
 * of each buffer abuts an unmapped 2 MiB range), UGVC_POISON=1|2 (new buffers filled with 0xA5 | 0xFF),
 * UGVC_DEBUG_SYNC=1 (every kernel launch named on stderr and waited for), UGVC_BREADCRUMB=1 (the last launches are
 * printed if the process aborts). */
int ugvc_selftest(ugvc_ctx* ctx, int64_t n);

/* ---- resident side tables --------------------------------------------------------------
 * ugvc_ref_upload: `--reference_file` (docs/filter_variants_pipeline.md:38-39); replaces the
 *   per-row pyfaidx fetches (pattern: ugvc/pipelines/vcfbed/calibrate_bridging_snvs.py:28-30).
 * ugvc_runs_upload: `--runs_file` + `--hpol_filter_length_dist L D` (docs/...md:30-33);
 *   runs shorter than min_len are dropped on upload; mark_hpol=0 keeps the two features but
 *   never sets UGVC_FLAG_HPOL_RUN.
 * ugvc_track_upload: one `--annotate_intervals` BED (docs/...md:45-46), track_id < UGVC_MAX_TRACKS.
 *   Interval tables (runs and tracks) must have non-decreasing starts AND ends inside every contig
 *   (sorted BED, nested intervals merged) - anything else is rejected with an error.
 * ugvc_blacklist_upload: `--blacklist` loci as sorted unique u64 keys contig<<32|pos (docs/...md:34-35).
 * ugvc_model_upload: one model of the `--model_file` dict entry `--model_name`, flattened
 *   (docs/...md:26-29; variantcalling_amd/model_io.py); one call per group.
 *   node i: feature[i] < 0 -> leaf with payload row left[i]; else left/right child indices.
 *   leaf_value: n_leaves x 2 doubles (RF: class fractions; GBT: margin, 0). */
int ugvc_ref_upload(ugvc_ctx* ctx, const uint8_t* codes, int64_t total_len,
                    const int64_t* contig_off, int n_contigs);
int ugvc_runs_upload(ugvc_ctx* ctx, const int32_t* starts, const int32_t* ends,
                     const int32_t* contig_ptr, int64_t n, int min_len, int max_dist, int mark_hpol);
int ugvc_track_upload(ugvc_ctx* ctx, int track_id, const int32_t* starts, const int32_t* ends,
                      const int32_t* contig_ptr, int64_t n);
int ugvc_set_n_tracks(ugvc_ctx* ctx, int n_tracks);
int ugvc_blacklist_upload(ugvc_ctx* ctx, const uint64_t* keys, int64_t n);
int ugvc_set_flow_order(ugvc_ctx* ctx, const char* flow4 /* e.g. "TGCA" */);
int ugvc_model_upload(ugvc_ctx* ctx, int group, int kind, const int32_t* feature,
                      const float* threshold, const int32_t* left, const int32_t* right,
                      int32_t n_nodes, const int32_t* tree_root, int32_t n_trees,
                      const double* leaf_value, int32_t n_leaves, int32_t n_features,
                      float base_score, int32_t max_depth);
/* No model for this variant type (a re-configured context forgets the previous one): its variants get
 * tree_score 0 and FILTER PASS, as when the reference's model dict has no entry for the type. */
int ugvc_model_clear(ugvc_ctx* ctx, int group);

/* ---- the hot path ----------------------------------------------------------------------
 * ugvc_filter_variants: annotate_concordance + blacklist apply + model predict + FILTER
 *   (call pattern ugvc/pipelines/run_no_gt_report.py:92-94,314).
 *   Synchronous for the caller; inside, a chunk pipeline (csrc/pipeline.hip): host threads check and pack row chunks
 *   into pinned slots while one DMA per chunk brings the previous one in, the scoring pass (one fused kernel per chunk;
 *   two launches when the indel forests are handed to a second kernel) runs on it and one DMA brings its three result
 *   columns back.  On return the callset and its results are resident, as after upload + ugvc_filter_resident.  On ANY
 *   error the context is left empty (n = 0).  Host threads: the pool and the calling thread move to the GPU's NUMA node
 *   for the duration of the call UNLESS the caller's affinity mask is already restricted (taskset, per-rank core
 *   binding), which is respected; UGVC_NO_NUMA=1 switches the placement off.
 * Resident form (inputs stay in HBM; used by bench.py and for pipelining):
 *   ugvc_variants_upload -> ugvc_filter_resident (async launch) -> ugvc_results_download.
 * ugvc_timed_filter: `iters` back-to-back launches bracketed by hipEvents on the context
 *   stream; returns total milliseconds (kernel time only, inputs resident).
 * ugvc_feature_matrix: the N x F float32 feature matrix (row-major) train_models_pipeline
 *   fits on (docs/train_models_pipeline.md:5-10); group[i] in {0,1,2} optional (may be NULL). */
int ugvc_filter_variants(ugvc_ctx* ctx, const ugvc_variants* v, const ugvc_results* out);
/* ugvc_reserve: everything ugvc_filter_variants allocates once per callset size (resident columns, pinned staging, streams,
 *   events, the worker pool) and the first use of the kernels' code object, without touching a row.  Optional: a tool that
 *   knows its callset's size early calls it from a helper thread beside its other set-up (reference / table / model uploads
 *   on another thread are fine; a pass is not) - the first ugvc_filter_variants over 5 M rows spends 44 of its 49 ms there.
 *   A callset that is resident survives the call only if no resident column has to grow for the reserved size; otherwise
 *   the context is left EMPTY (n = 0: ugvc_filter_resident / ugvc_results_download / ugvc_sec_apply see no rows).
 *   Replaces nothing in the reference (its loops allocate as they go: ugvc/pipelines/vcfbed/calibrate_bridging_snvs.py:101-130). */
int ugvc_reserve(ugvc_ctx* ctx, int64_t n_variants, int64_t alleles_len);
/* What is resident: the row count of the callset ugvc_filter_resident / ugvc_results_download / ugvc_sec_apply would work on
 * (0 after any failed upload and after a reservation that had to re-allocate a resident column) and whether the result columns
 * hold a scoring pass over it.  Either pointer may be NULL.  (Builder-defined; the reference keeps its callset in a DataFrame.) */
int ugvc_resident_count(ugvc_ctx* ctx, int64_t* n_variants, int* scored);
int ugvc_variants_upload(ugvc_ctx* ctx, const ugvc_variants* v);
int ugvc_filter_resident(ugvc_ctx* ctx);
int ugvc_results_download(ugvc_ctx* ctx, const ugvc_results* out);
int ugvc_timed_filter(ugvc_ctx* ctx, int iters, float* ms_total);
/* `iters` steps of {filter kernel [+ all-gather of the result columns when gather != 0]} on the
 * context stream; ms_total = first-to-last event, ms_kernel = sum of the per-launch kernel
 * event pairs (what bench.py's roofline figure divides by). */
int ugvc_timed_steps(ugvc_ctx* ctx, int iters, int64_t shard_cap, int gather, float* ms_total,
                     float* ms_kernel);
/* on = 1 (default): ugvc_timed_steps brackets EVERY step with its own event pair (per-step times: ugvc_last_step_ms);
 * on = 0: one pair around the whole run, the passes back to back as a production stream issues them - an event record is a
 * marker the stream drains to, ~9 us per step; ms_kernel is then the time between that pair (measurement knob, builder-defined) */
int ugvc_set_step_events(ugvc_ctx* ctx, int on);
/* The shader clock the resident scoring pass sustains (what bench.py prices its instruction-issue floor at, instead of the
 * device's peak clock): `passes` passes back to back, the last one probed by two counters inside the kernel - the constant
 * 100 MHz s_memrealtime and the shader-clock s_memtime at the entry and the end of one wave.  wave_ms (nullable): that wave's
 * span in milliseconds.  Builder-defined measurement aid. */
int ugvc_pass_clock(ugvc_ctx* ctx, int passes, double* shader_ghz, double* wave_ms);
/* per-step kernel milliseconds of the last ugvc_timed_steps (HIP event pairs on the launch stream): copies at most
 * `cap` values, returns how many (bench.py's p5 / p95) */
int ugvc_last_step_ms(ugvc_ctx* ctx, float* out, int cap);
int ugvc_device_sync(ugvc_ctx* ctx);   /* hipDeviceSynchronize on the context's device */
int ugvc_feature_matrix(ugvc_ctx* ctx, float* x_host, uint8_t* group_host);
/* `iters` back-to-back builds of the resident N x F matrix, nothing downloaded: total milliseconds on the context stream
 * (bench.py --workload c5_gemm reports the build in GB/s of its algorithmic bytes: 121.6 read + 4 F written per variant) */
int ugvc_timed_feature_matrix(ugvc_ctx* ctx, int iters, float* ms_total);
int ugvc_n_features(ugvc_ctx* ctx);
/* Config C5 (train_models_pipeline: on-GPU feature matrix + tree-ensemble inference as a leaf-matrix
 * GEMM on the matrix cores; docs/train_models_pipeline.md:46-54 `--evaluate_concordance`): evaluate the
 * uploaded additive (UGVC_MODEL_GBT, depth <= 6) ensemble of `group` on rows of the RESIDENT feature
 * matrix left by ugvc_feature_matrix.  rows == NULL: every row.  use_mfma 1: path-matrix GEMM
 * (v_mfma_i32_16x16x64_i8), 0: row traversal - identical f32 margins.  ms_per_launch (optional): mean
 * kernel time of `iters` launches (HIP events). */
int ugvc_forest_gemm(ugvc_ctx* ctx, int group, const int32_t* rows, int64_t n_rows, int use_mfma, int iters,
                     float* margin_out, float* ms_per_launch);
/* The three variant-type groups in ONE launch (round 5): rows[k] / n_rows[k] = group k's rows of the resident matrix (n_rows[k]
 * may be 0), every group with rows holding an additive depth <= 6 ensemble of the same kind; margin_out[row] (N floats, by row of
 * the resident matrix) receives each named row's margin, other entries are left as they are.  Identical f32 margins to the
 * traversal (same compares, same additions in tree order). */
int ugvc_forest_gemm3(ugvc_ctx* ctx, const int32_t* const* rows, const int64_t* n_rows, int iters, float* margin_out,
                      float* ms_per_launch);
/* Host-only (no GPU): the single-base-substitution cycle-skip table the kernels use for a flow
 * order; index = (last left base)<<6 | ref<<4 | alt<<2 | (first right base), bases A,C,G,T = 0..3,
 * value 0 non-skip / 1 possible-cycle-skip / 2 cycle-skip.  Exposed so CPU tests can check it
 * against the full flow-key computation. */
int ugvc_host_css_lut(const char* flow4, uint8_t out[256]);
/* Kernel-path selection and profiling knobs (A/B measurements, parity cross-checks; 0 = production path):
 *   bits 0-4 section ablation of the featurize kernel (results are WRONG): 1 no forest kernel, 2 no joins,
 *            4 no quantisation, 8 no window features, 16 no record append
 *   32   featurize kernel without the one-tile-ahead column prefetch     64   phase clocks on (ugvc_debug_phase_clocks)
 *   256  v1 universal fused kernel                                       1024 pair-sum forest kernel (v3)
 *   2048 16 trees in flight in the forest kernel (v3)                    65536 v3 kernels instead of v5
 *   bits 12-13 v3 featurize workgroups per CU (1..3; 0 = 4)              bits 14-15 v3 forest waves (1: 12, 2: 8, 3: 4; 0 = 16)
 *   v5 profiling (results WRONG / partial): 131072 no SNP walk, 262144 no indel tiles, 524288 no side-table joins;
 *   (results unchanged) bits 24-27: waves per workgroup that may take indel tiles (0 = a quarter), bit 28: 8 instead of
 *   16 trees in flight (3-track kernel), bit 29: no s_setprio */
int ugvc_set_kernel_variant(ugvc_ctx* ctx, int variant);
/* Profiling aid: core-clock cycles wave 0 of every featurize workgroup spent between the kernel's phase
 * boundaries (kernel variant bit 6 turns the clocks on), summed over workgroups and launches since the
 * last reset; out[6] = tiles counted. */
int ugvc_debug_phase_clocks(ugvc_ctx* ctx, uint64_t out[8], int reset);

/* ---- score evaluation on the GPU ----------------------------------
 * ugvc_eval_counts: integer counts under the accuracy table of `train_models_pipeline --evaluate_concordance`
 * (calc_accuracy_metrics: ugvc/pipelines/evaluate_concordance.py:27,100; group names
 * test/resources/system/test_evaluate_concordance/expected.out.stats.csv:1-10) on the RESIDENT FILTER column of the
 * last scoring pass.  label[i]: 1 true call, 0 false call, <0 unlabelled; cat_bits[i]: bit c set = row belongs to
 * category c (<= 16); out[c] = {true, false, true & PASS, false & PASS}.
 * ugvc_pr_curve: cumulative recall / precision / f1 curve of ReportUtils.__calc_performance
 * (ugvc/reports/report_utils.py:494-504; formulas ugvc/utils/stats_utils.py:76-138): rows sorted by score
 * ascending (stable; NaN last), cls[i] = 1 tp / 2 fp / 0 other; per position c_fn = initial_fn + cumsum(tp),
 * c_tp = initial_tp - cumsum(tp), c_fp = initial_fp - cumsum(fp).  Outputs have n entries; order (nullable) = the
 * permutation; ms_device (nullable) = device time of sort + scans + finish. */
int ugvc_eval_counts(ugvc_ctx* ctx, const int8_t* label, const uint16_t* cat_bits, int64_t out[16][4]);
int ugvc_pr_curve(ugvc_ctx* ctx, const double* score, const uint8_t* cls, int64_t n, int64_t initial_tp, int64_t initial_fp,
                  int64_t initial_fn, double* sorted_score, double* recall, double* precision, double* f1, int32_t* order,
                  float* ms_device);

/* ---- pileup allele/strand/base-quality tally (builder-defined) ---------
 * offsets: n_loci+1 CSR offsets into obs; obs u16 = allele(2b: 0 ref,1 alt,2 other) |
 * strand<<2 | bq<<3. */
int ugvc_pileup_tally(ugvc_ctx* ctx, const int64_t* offsets, const uint16_t* obs,
                      int64_t n_loci, const ugvc_pileup_out* out);
int ugvc_pileup_upload(ugvc_ctx* ctx, const int64_t* offsets, const uint16_t* obs, int64_t n_loci);
int ugvc_timed_pileup(ugvc_ctx* ctx, int iters, float* ms_total);

/* ---- SEC noisy-locus statistic (/root/reference/ugvc/utils/stats_utils.py:12-70) -------
 * per locus: k categories of observed and expected counts -> multinomial likelihood of the
 * observation under the add-one corrected expected frequencies, and its ratio to the
 * likelihood under the observation's own frequencies. */
int ugvc_sec_likelihood_ratio(ugvc_ctx* ctx, const int32_t* actual, const int32_t* expected,
                              int64_t n_loci, int k, double* likelihood, double* ratio);

/* ---- SEC database: build + apply.  The tools (sec_training,
 * correct_systematic_errors: /root/reference/ugvc/__main__.py:19,56) are absent and undocumented (README.md:12); the
 * statistic is in-tree (stats_utils.py:12-70) and a call the cohort's noise explains is tagged "SEC"
 * (ugvc/reports/report_utils.py:71-75).  BUILDER-DEFINED around those two facts:
 *   build : n_obs (key, k counts) observations of a cohort, any order, keys = contig << 32 | pos  ->  sorted unique
 *           keys + per-locus summed counts (out arrays sized n_obs by the caller; *out_n = loci);
 *   upload: the database a context applies (keys sorted, unique; k = 2: ref, alt; k = 3: ref, alt, other);
 *   apply : per RESIDENT variant on a database locus: observed = (ad_ref, ad_alt[, max(dp - ad_ref - ad_alt, 0)]),
 *           expected rescaled to the observed depth with scale_contingency_table when scale_expected != 0,
 *           ratio[i] = multinomial_likelihood_ratio(observed, expected)[1], is_sec[i] = ratio >= min_ratio; off the
 *           database ratio = NaN, is_sec = 0.  mark != 0 also ORs UGVC_FLAG_SEC into the resident flags column (after
 *           ugvc_filter_resident), which the gather / download then carry.  ratio / is_sec may be NULL. */
int ugvc_sec_db_build(ugvc_ctx* ctx, const uint64_t* keys, const int32_t* counts, int64_t n_obs, int k,
                      uint64_t* out_keys, int32_t* out_expected, int64_t* out_n);
int ugvc_sec_db_upload(ugvc_ctx* ctx, const uint64_t* keys, const int32_t* expected, int64_t n_db, int k);
int ugvc_sec_apply(ugvc_ctx* ctx, double min_ratio, int scale_expected, int mark, double* ratio, uint8_t* is_sec);
/* `iters` mark-only applications (as ugvc_sec_apply(..., mark = 1, NULL, NULL)) back to back between two events on the
 * context stream: total milliseconds of kernel time. */
int ugvc_timed_sec_apply(ugvc_ctx* ctx, double min_ratio, int scale_expected, int iters, float* ms_total);

/* ---- is_homopolymer_snp + VAF gate (/root/reference/ugvc/pipelines/vcfbed/
 * calibrate_bridging_snvs.py:9-66,110-126): out_pass[i]=1 when the record is un-filtered. */
typedef struct ugvc_bridging_params {
    double min_initial_qual; double min_tumor_vaf; double max_normal_vaf;
    int min_query_hmer_size; int min_normal_depth; int min_distance_from_edge;
} ugvc_bridging_params;
int ugvc_bridging_snvs(ugvc_ctx* ctx, const ugvc_variants* v, const uint8_t* is_pass,
                       const int32_t* ad_alt_sum, const int32_t* bg_ad_alt_sum, const int32_t* bg_dp,
                       const ugvc_bridging_params* p, uint8_t* out_hmer_snp, uint8_t* out_pass);

/* ---- multi-GPU: reassemble the scored callset with an RCCL all-gather over xGMI ---------
 * One process per GPU.  Rank 0 calls ugvc_comm_unique_id and shares the 128 bytes out of
 * band; every rank calls ugvc_comm_init.  ugvc_allgather_results gathers the resident
 * result columns of every rank (shards padded to shard_cap rows) into host or device
 * order-preserving buffers: rank r's rows land at [r*shard_cap, r*shard_cap + counts[r]).
 * The collective is issued on a second HIP stream behind an event, so the gather of one scoring pass
 * overlaps the kernels of the next (ugvc_timed_steps, streaming use); ugvc_gathered_download and
 * ugvc_gather_fence order it before anything that reads or rewrites the gather buffers. */
int ugvc_comm_unique_id(uint8_t id[128]);
int ugvc_comm_init(ugvc_ctx* ctx, const uint8_t id[128], int rank, int world);
int ugvc_comm_destroy(ugvc_ctx* ctx);
/* what RCCL itself reports about the communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice): bench.py and the
 * multi-rank tool print it, so a run that silently fell back to one rank cannot pass for N */
int ugvc_comm_info(ugvc_ctx* ctx, int* nranks, int* rank, int* device);
int ugvc_allgather_resident(ugvc_ctx* ctx, int64_t shard_cap);   /* async: collective on its own stream */
int ugvc_gather_fence(ugvc_ctx* ctx);                             /* context stream waits for the outstanding gathers */
/* zero-copy form used by ugvc_timed_steps: ugvc_gather_target picks the next of two gather buffers (waiting for
 * its previous collective) and returns this rank's slot so the scoring pass writes there; ugvc_gather_launch
 * issues the collective behind the work queued on the context stream. */
int ugvc_gather_target(ugvc_ctx* ctx, int64_t shard_cap, float** score, uint8_t** filter, uint8_t** flags);
int ugvc_gather_launch(ugvc_ctx* ctx, int64_t shard_cap);
int ugvc_gathered_download(ugvc_ctx* ctx, int64_t shard_cap, int world, const ugvc_results* out);

#ifdef __cplusplus
}
#endif
#endif
