"""CPU ORACLE - TEST INFRASTRUCTURE ONLY.  The encodings the oracle computes in, restated HERE with their evidence.

The product keeps its own copy of these numbers (`variantcalling_amd/schema.py`, `include/ugvc_mi355x.h`).  The oracle
must not read that copy - a slip there would cancel out in every oracle-vs-GPU comparison - so feature order, class
codes, output bits and model kinds are written down a second time below, each with the reference line it follows;
`tests/test_oracle_golden.py::test_oracle_spec_matches_the_product_schema` is the one place the two are compared.
Only the plain containers (`schema.Reference`, `IntervalTrack`, `VariantTable`, `FlatForest`, `FilterResult`: named
arrays, no numbers) are shared with the product.  Citations are relative to /root/reference.
"""
from __future__ import annotations

import numpy as np

# ---- base alphabet: the RTG SDF fixtures of the reference store N, A, C, G, T as 0..4
# (test/resources/general/chr1_head/Homo_sapiens_assembly38.fasta.sdf)
BASE_N, BASE_A, BASE_C, BASE_G, BASE_T = 0, 1, 2, 3, 4
CODE_TO_CHAR = "NACGT"
_ASCII = np.zeros(256, dtype=np.uint8)
for _i, _ch in enumerate(CODE_TO_CHAR):
    _ASCII[ord(_ch)] = _i
    _ASCII[ord(_ch.lower())] = _i


def encode_bases(seq) -> np.ndarray:
    """ASCII bases -> codes; anything that is not ACGT / acgt is N."""
    if isinstance(seq, str):
        seq = seq.encode()
    return _ASCII[np.frombuffer(seq, dtype=np.uint8)]


def decode_bases(codes) -> str:
    return "".join(CODE_TO_CHAR[int(c)] for c in codes)


# ---- features
MOTIF_SIZE = 5            # get_motif_around(df, 5, fasta): ugvc/pipelines/run_no_gt_report.py:94
GC_WINDOW = 10            # gc_content window: BUILDER-DEFINED (the body is in the absent submodule)
# Order of the model's input columns.  The columns themselves are the reference's: qual / sor / dp / ad / gq come from
# get_vcf_df (ugvc/reports/report_wo_gt.ipynb:1207-1210; header fields test/resources/unit/vcfbed/test_vcftools/
# header.txt:3379,3391-3398), vaf = ad / dp (ugvc/reports/report_data_loader.py:24-28), the annotate_concordance columns
# indel_classify, indel_length, hmer_indel_length, hmer_indel_nuc, left_motif, right_motif (ugvc/pipelines/
# run_no_gt_report.py:133-143), gc_content and hmer_indel_length (ugvc/reports/report_data_loader.py:67-92),
# cycleskip_status (header.txt:3382 X_CSS), the hpol-run flags behind HPOL_RUN (docs/filter_variants_pipeline.md:30-33)
# and one boolean per --annotate_intervals file (report_data_loader.py:94).  Their ORDER is BUILDER-DEFINED (an estimator
# fitted on a named frame is re-indexed by name: model_io.feature_permutation).
BASE_FEATURES = (
    "qual", "sor", "dp", "ad_ref", "ad_alt", "vaf", "gq",
    "indel_classify", "indel_length", "hmer_indel_length", "hmer_indel_nuc",
    "left_motif", "right_motif", "gc_content", "cycleskip_status",
    "inside_hmer_run", "close_to_hmer_run",
)
N_BASE_FEATURES = 17
MAX_TRACKS = 5

# indel_classify: the reference's None / 'ins' / 'del' (ugvc/pipelines/run_no_gt_report.py:133-143) as 0 / 1 / 2
INDEL_NONE, INDEL_INS, INDEL_DEL = 0, 1, 2
# cycleskip_status: X_CSS in {non-skip, possible-cycle-skip, cycle-skip} (header.txt:3382), NA for indels
CSS_NON_SKIP, CSS_POSSIBLE, CSS_CYCLE_SKIP, CSS_NA = 0, 1, 2, 3
CSS_NAMES = ("non-skip", "possible-cycle-skip", "cycle-skip", "NA")
# one model per variant type: VARIANT_TYPE in {snp, h-indel, non-h-indel} (header.txt:3381; category bins
# ugvc/reports/report_utils.py:508-538)
GROUP_SNP, GROUP_HINDEL, GROUP_NON_HINDEL = 0, 1, 2
GROUP_NAMES = ("snp", "h-indel", "non-h-indel")
N_GROUPS = 3

# ---- outputs: FILTER PASS | LOW_SCORE, INFO TREE_SCORE, tags HPOL_RUN / COHORT_FP (docs/howto-callset-filter.md:61-65;
# HPOL_RUN ignored by the evaluation: ugvc/pipelines/evaluate_concordance.py:44-48), SEC (ugvc/reports/report_utils.py:
# 71-75), interval columns (report_data_loader.py:94).  The BIT each tag travels in is BUILDER-DEFINED.
FILTER_PASS, FILTER_LOW_SCORE = 0, 1
FLAG_HPOL_RUN, FLAG_COHORT_FP, FLAG_SEC = 1, 2, 4
FLAG_TRACK0_SHIFT = 3

# ---- model kinds: scikit-learn forest (f32 feature <= f64 threshold, mean of the trees' class fractions in f64:
# setup/environment.yml:399) / XGBoost (f32 feature < f32 threshold, f32 additive margin, sigmoid: environment.yml:354)
MODEL_RF, MODEL_GBT = 0, 1
