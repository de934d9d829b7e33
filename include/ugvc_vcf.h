/* ugvc_vcf.h - C ABI of the native VCF <-> SoA codec (host side of the hot path).
 *
 * Replaces the two per-record pysam loops that bracket featurize -> score -> FILTER in the reference:
 *   read : ugbio_core.vcfbed.vcftools.get_vcf_df  (body absent; call sites
 *          /root/reference/ugvc/pipelines/run_no_gt_report.py:307-312, shape quoted in
 *          /root/reference/ugvc/reports/report_wo_gt.ipynb:1207-1210; field dictionary
 *          /root/reference/test/resources/unit/vcfbed/test_vcftools/header.txt:3369-3398)
 *   write: header add + per-record filter.add / info[...] = (in-tree instance of the pattern:
 *          /root/reference/ugvc/pipelines/vcfbed/calibrate_bridging_snvs.py:101-130; tags
 *          /root/reference/docs/howto-callset-filter.md:65, ugvc/pipelines/evaluate_concordance.py:47)
 *
 * The library (libugvc_vcf.so) has no GPU dependency: BGZF blocks are inflated / deflated and records are
 * tokenised / spliced on host threads; its output columns are exactly the `ugvc_variants` table of
 * ugvc_mi355x.h.  Semantics are those of variantcalling_amd/io/vcf.py (the pure-Python host reference the
 * parity tests compare against, byte for byte): multi-allelic records are featurised on their first ALT,
 * missing numeric fields read as 0, rows are stably sorted by (contig, pos).
 *
 * Conventions: 0 on success, <0 on error with ugvc_vcf_last_error() (thread-local); the handle owns every
 * array a view points to until ugvc_vcf_free.
 */
#ifndef UGVC_VCF_H
#define UGVC_VCF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ugvc_vcf ugvc_vcf;

typedef struct ugvc_vcf_view {
    int64_t n;                 /* records == table rows                                               */
    int64_t pool_bytes;        /* length of `alleles`                                                  */
    /* ugvc_variants columns, rows sorted by (contig, pos) */
    const uint16_t* contig;
    const int32_t* pos;
    const uint16_t* ref_len;
    const uint16_t* alt_len;
    const uint32_t* ref_off;
    const uint32_t* alt_off;
    const uint8_t* alleles;    /* base codes N=0 A=1 C=2 G=3 T=4, REF then first ALT of every row      */
    const float* qual;         /* QUAL, or 10 * max(INFO/TLOD) in mutect mode                          */
    const float* sor;          /* INFO/SOR                                                             */
    const int32_t* dp;         /* FORMAT/DP, AD[0], AD[1] of the chosen sample                         */
    const int32_t* ad_ref;
    const int32_t* ad_alt;
    const uint8_t* gq;
    const uint8_t* gt;         /* 0 hom-ref / no call, 1 het (any allele 1), 2 hom-alt "1/1"           */
    const float* tlod;         /* max(INFO/TLOD)                                                       */
    const uint8_t* has_id;     /* ID column != "."                                                     */
    const int64_t* order;      /* table row k is record order[k] of the file                           */
    /* text */
    const char* header;        /* header lines (every line starting with '#'), joined by '\n'          */
    int64_t header_bytes;
    const char* text;          /* inflated file; FILTER column of table row k = text[filter_off[k] ..] */
    const int64_t* filter_off;
    const int32_t* filter_len;
    /* multi-allelic records: the table row carries the FIRST ALT; the host expands the rows with n_alt > 1 (or an ALT
     * of '*') from the record text: table row k is text[rec_off[k] .. rec_off[k] + rec_len[k]) */
    const uint8_t* n_alt;      /* ALT alleles of the record (capped at 255)                            */
    const int64_t* rec_off;
    const int32_t* rec_len;
} ugvc_vcf_view;

/* Read + inflate (.gz: BGZF blocks in parallel, plain gzip serially) + tokenise `path`.
 * contig_names: the reference's contig order (index = `contig` column).  sample: 0-based sample column.
 * n_threads <= 0: one per hardware thread (capped at 64). */
int ugvc_vcf_read(const char* path, const char* const* contig_names, int n_contigs, int is_mutect, int sample,
                  int n_threads, ugvc_vcf** out);
/* One PART of the callset, for one rank of a multi-process run (docs/run_comparison_pipeline.md:81 is the reference's own
 * multi-process idiom): the file is inflated and cut into lines as above, then only records [R part / n_parts, R (part + 1) /
 * n_parts) IN FILE ORDER are tokenised, ordered and turned into columns - the equal-count shard of that rank when the file is
 * sorted by (contig, pos), which the caller has to establish across ranks (first / last key and sortedness of every part).
 * The handle cannot be written back (ugvc_vcf_write_filtered needs the whole file).  ugvc_vcf_part_info: records in the
 * file and the file-order index of this part's first record. */
int ugvc_vcf_read_part(const char* path, const char* const* contig_names, int n_contigs, int is_mutect, int sample,
                       int n_threads, int part, int n_parts, ugvc_vcf** out);
int ugvc_vcf_part_info(const ugvc_vcf* h, int64_t* n_total, int64_t* part_lo);
/* A function left here is called ONCE - on the calling thread, by the next ugvc_vcf_read / ugvc_vcf_read_part of THIS thread, as
 * soon as the record lines are counted (a third of the way through the read) - and forgotten: a tool prepares what depends on
 * the callset's size meanwhile (filter_variants_pipeline: ugvc_reserve of the GPU engine).  NULL clears it.  (The reference's
 * reader has no such point: it yields records one by one, report_wo_gt.ipynb:1207-1210.) */
void ugvc_vcf_set_count_hook(void (*fn)(int64_t n_records, int64_t text_bytes, void* user), void* user);
int ugvc_vcf_get_view(const ugvc_vcf* h, ugvc_vcf_view* view);

/* Write the input records in their original order with FILTER := PASS | [HPOL_RUN;][COHORT_FP;][LOW_SCORE],
 * INFO += TREE_SCORE=<shortest round-trip f32>[;HPOL_RUN] (older TREE_SCORE / HPOL_RUN entries dropped) and
 * the five header lines if absent.  Result columns are in TABLE order (length n); cohort_extra (nullable)
 * marks additional COHORT_FP rows.  `out_path` ending in ".gz" is written as BGZF (65280-byte blocks,
 * zlib level 6, EOF marker). */
int ugvc_vcf_write_filtered(const ugvc_vcf* h, const char* out_path, const float* tree_score, const uint8_t* filter,
                            const uint8_t* flags, const uint8_t* cohort_extra, int64_t n, int n_threads, int write_index);
/* write_index != 0 and a ".gz" path: `out_path`.tbi is written too (tabix index, format VCF; the reference ends its
 * write loop with pysam.tabix_index, ugvc/pipelines/vcfbed/calibrate_bridging_snvs.py:130).  Returns 1 (file written,
 * no index) when the records are not grouped by contig and sorted by position - tabix refuses such files as well. */

void ugvc_vcf_free(ugvc_vcf* h);
const char* ugvc_vcf_last_error(void);
int ugvc_vcf_abi_version(void);
/* Which deflate implementation the BGZF reader / writer use: 0 = automatic (libdeflate.so.0 when the host has it - 2-3 x
 * zlib's speed on 64 KB blocks - else zlib; the environment variable UGVC_DEFLATE=zlib means 1), 1 = zlib level 6 (the
 * compressed BYTES are then those of the pure-Python reference codec io/vcf.py), 2 = libdeflate level 6 (fails if absent).
 * Returns the implementation now in use (1 or 2), or -1 with ugvc_vcf_last_error() set.  Inflated text, CRCs, block structure
 * and the meaning of the tabix index are the same either way. */
int ugvc_vcf_set_deflate(int backend);

/* ---- side tables of the same tools: FASTA -> base codes, BED / interval_list -> interval arrays ----------------
 * `--reference_file` is opened with pyfaidx and fetched per row in the reference
 * (/root/reference/ugvc/pipelines/vcfbed/calibrate_bridging_snvs.py:28-30); here the whole FASTA (plain, gzip or
 * BGZF) is encoded once: record name = first token after '>', sequence bytes mapped A/a C/c G/g T/t -> 1..4,
 * everything else 0, line ends dropped.  `--runs_file` / `--annotate_intervals`
 * (/root/reference/docs/filter_variants_pipeline.md:30-33,45-46): one (contig index, start, end) row per data line
 * whose contig is in `contig_names`, 0-based half-open; interval_list files (an '@' header line, or the file
 * extension) are 1-based inclusive and converted.  ugvc_intervals_track sorts / merges.  Semantics are those
 * of variantcalling_amd/io/fasta.py and bed.py (tests/test_vcf_native.py compares). */
typedef struct ugvc_fasta ugvc_fasta;
typedef struct ugvc_fasta_view {
    int64_t total;               /* bases over all records                                   */
    int32_t n_contigs;
    const uint8_t* codes;        /* total bytes                                              */
    const int64_t* contig_off;   /* n_contigs + 1                                            */
    const char* names;           /* record names joined by '\n'                              */
    int64_t names_bytes;
} ugvc_fasta_view;
int ugvc_fasta_read(const char* path, int n_threads, ugvc_fasta** out);
int ugvc_fasta_get_view(const ugvc_fasta* h, ugvc_fasta_view* view);
void ugvc_fasta_free(ugvc_fasta* h);

typedef struct ugvc_intervals ugvc_intervals;
typedef struct ugvc_intervals_view {
    int64_t n;
    const int64_t* contig;       /* index into contig_names                                  */
    const int64_t* start;        /* 0-based                                                  */
    const int64_t* end;          /* exclusive                                                */
} ugvc_intervals_view;
int ugvc_intervals_read(const char* path, const char* const* contig_names, int n_contigs, int n_threads,
                        ugvc_intervals** out);
int ugvc_intervals_get_view(const ugvc_intervals* h, ugvc_intervals_view* view);
/* The finished track of the rows read (round 6; replaces the numpy post-processing of variantcalling_amd/io/bed.py:
 * track_from_arrays, which stays as its checker): rows ordered by (contig, start, end), rows with end <= start dropped,
 * with `merge` != 0 overlapping / nested / book-ended rows of a contig folded into one (a new interval opens where a start
 * lies beyond every end seen so far), 32-bit coordinates, ptr[c] .. ptr[c + 1] = the rows of contig c (n_contigs + 1 entries)
 * - the layout ugvc_runs_upload / ugvc_track_upload take (include/ugvc_mi355x.h).  The view lives as long as the handle. */
typedef struct ugvc_track_view {
    int64_t n;
    const int32_t* start;
    const int32_t* end;
    const int32_t* ptr;
} ugvc_track_view;
int ugvc_intervals_track(ugvc_intervals* h, int n_contigs, int merge, ugvc_track_view* view);
void ugvc_intervals_free(ugvc_intervals* h);

/* The codec's file reader, bare: a plain / gzip / BGZF file -> its (inflated) bytes, BGZF members one block per thread through
 * the selected deflate back end (ugvc_vcf_set_deflate) - what htslib's bgzf_read delivers.  For tools and for the parity test on
 * a real htslib stream (tests/test_htslib_bgzf.py).  `*data` lives until ugvc_blob_free. */
typedef struct ugvc_blob ugvc_blob;
int ugvc_bgzf_read(const char* path, int n_threads, ugvc_blob** out, const char** data, int64_t* len);
void ugvc_blob_free(ugvc_blob* b);

/* Shortest decimal string that round-trips the f32 (fixed notation, at least one fractional digit):
 * the TREE_SCORE formatter, exposed for the parity test against numpy.format_float_positional. */
int ugvc_vcf_format_f32(float x, char* buf, int cap);

#ifdef __cplusplus
}
#endif
#endif
