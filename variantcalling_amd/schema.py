"""Column-major (SoA) tables that cross the C ABI, and the encodings both sides agree on.

Everything here is plain numpy: each column is one contiguous array, variants are sorted
by (contig, pos).  The layout is the one fixed in SURVEY.md §8(d) / BASELINE.md (the
"algorithmic bytes per variant" accounting uses exactly these widths).

Reference-side meaning of the columns (the reference's own table is a pandas DataFrame
built by `ugbio_core.vcfbed.vcftools.get_vcf_df`; call sites
ugvc/pipelines/run_no_gt_report.py:307-312, shape quoted in
ugvc/reports/report_wo_gt.ipynb:1207-1210):

    chrom,pos,ref,alleles -> contig,pos,ref_off/ref_len,alt_off/alt_len (+ allele pool)
    qual, sor, dp, ad, gq, gt -> qual, sor, dp, ad_ref/ad_alt, gq, gt
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

# ---- base alphabet: identical to the RTG SDF fixture of the reference
BASE_N, BASE_A, BASE_C, BASE_G, BASE_T = 0, 1, 2, 3, 4
CODE_TO_CHAR = "NACGT"
_ASCII_TO_CODE = np.zeros(256, dtype=np.uint8)
for _i, _ch in enumerate(CODE_TO_CHAR):
    _ASCII_TO_CODE[ord(_ch)] = _i
    _ASCII_TO_CODE[ord(_ch.lower())] = _i


def encode_bases(seq: str | bytes) -> np.ndarray:
    """ASCII bases -> u8 codes (anything that is not ACGT/acgt becomes N=0)."""
    if isinstance(seq, str):
        seq = seq.encode()
    return _ASCII_TO_CODE[np.frombuffer(seq, dtype=np.uint8)]


def decode_bases(codes) -> str:
    return "".join(CODE_TO_CHAR[int(c)] for c in codes)


# ---- feature vector handed to the tree ensembles (order is part of the ABI: ugvc_mi355x.h)
MOTIF_SIZE = 5          # get_motif_around(df, 5, fasta): ugvc/pipelines/run_no_gt_report.py:94
GC_WINDOW = 10          # gc_content window (builder-defined)
BASE_FEATURES = (
    "qual", "sor", "dp", "ad_ref", "ad_alt", "vaf", "gq",
    "indel_classify", "indel_length", "hmer_indel_length", "hmer_indel_nuc",
    "left_motif", "right_motif", "gc_content", "cycleskip_status",
    "inside_hmer_run", "close_to_hmer_run",
)
N_BASE_FEATURES = len(BASE_FEATURES)  # 17; + one boolean per annotation track
MAX_TRACKS = 5                        # flags bits 3..7

# indel_classify codes (reference strings None/'ins'/'del': run_no_gt_report.py:133-143)
INDEL_NONE, INDEL_INS, INDEL_DEL = 0, 1, 2
# cycleskip_status codes (X_CSS strings, test/resources/unit/vcfbed/test_vcftools/header.txt:3382)
CSS_NON_SKIP, CSS_POSSIBLE, CSS_CYCLE_SKIP, CSS_NA = 0, 1, 2, 3
CSS_NAMES = ("non-skip", "possible-cycle-skip", "cycle-skip", "NA")
# variant-type groups, one model each (VARIANT_TYPE header.txt:3381; report_utils.py:508-538)
GROUP_SNP, GROUP_HINDEL, GROUP_NON_HINDEL = 0, 1, 2
GROUP_NAMES = ("snp", "h-indel", "non-h-indel")
N_GROUPS = 3

# output columns
FILTER_PASS, FILTER_LOW_SCORE = 0, 1
FLAG_HPOL_RUN, FLAG_COHORT_FP, FLAG_SEC = 1, 2, 4
FLAG_TRACK0_SHIFT = 3

# model kinds
MODEL_RF = 0      # sklearn forest: f32 feature <= threshold, mean of f64 leaf class fractions
MODEL_GBT = 1     # XGBoost-style: f32 feature < threshold, f32 additive margin, sigmoid


def feature_names(n_tracks: int) -> tuple[str, ...]:
    return BASE_FEATURES + tuple(f"track{t}" for t in range(n_tracks))


@dataclass
class Reference:
    """Concatenated reference genome, 1 byte per base (codes 0..4)."""
    codes: np.ndarray                   # u8 [total]
    contig_off: np.ndarray              # i64 [n_contigs + 1]
    names: list[str] = field(default_factory=list)

    @property
    def n_contigs(self) -> int:
        return int(self.contig_off.size - 1)

    def contig_len(self, c: int) -> int:
        return int(self.contig_off[c + 1] - self.contig_off[c])


@dataclass
class IntervalTrack:
    """Sorted, non-overlapping BED intervals of one track, CSR by contig.

    Coordinates are kept exactly as they stand in the BED file (0-based start, exclusive
    end) and compared with the 1-based VCF POS the way the reference does it."""
    starts: np.ndarray                  # i32 [n]
    ends: np.ndarray                    # i32 [n]
    contig_ptr: np.ndarray              # i32 [n_contigs + 1]
    name: str = ""


@dataclass
class VariantTable:
    contig: np.ndarray                  # u8
    pos: np.ndarray                     # i32, 1-based
    ref_len: np.ndarray                 # u16
    alt_len: np.ndarray                 # u16
    ref_off: np.ndarray                 # u32 into alleles
    alt_off: np.ndarray                 # u32 into alleles
    alleles: np.ndarray                 # u8 pool of base codes
    qual: np.ndarray                    # f32
    sor: np.ndarray                     # f32
    dp: np.ndarray                      # i32
    ad_ref: np.ndarray                  # i32
    ad_alt: np.ndarray                  # i32
    gq: np.ndarray                      # u8
    gt: np.ndarray                      # u8 (0 other, 1 het, 2 hom-alt); host-only column

    @property
    def n(self) -> int:
        return int(self.pos.size)

    COLS = ("contig", "pos", "ref_len", "alt_len", "ref_off", "alt_off",
            "qual", "sor", "dp", "ad_ref", "ad_alt", "gq", "gt")
    DTYPES = dict(contig=np.uint16, pos=np.int32, ref_len=np.uint16, alt_len=np.uint16,
                  ref_off=np.uint32, alt_off=np.uint32, qual=np.float32, sor=np.float32,
                  dp=np.int32, ad_ref=np.int32, ad_alt=np.int32, gq=np.uint8, gt=np.uint8)

    def validate(self) -> None:
        n = self.n
        for c in self.COLS:
            a = getattr(self, c)
            if a.dtype != self.DTYPES[c] or a.shape != (n,) or not a.flags.c_contiguous:
                raise ValueError(f"column {c}: want contiguous {self.DTYPES[c].__name__}[{n}], "
                                 f"got {a.dtype}{a.shape}")
        if self.alleles.dtype != np.uint8:
            raise ValueError("allele pool must be u8")
        if n:
            key = (self.contig.astype(np.int64) << 32) | self.pos.astype(np.int64)
            if np.any(key[1:] < key[:-1]):
                raise ValueError("variants must be sorted by (contig, pos)")
            if np.any(self.ref_len == 0) or np.any(self.alt_len == 0):
                raise ValueError("empty alleles are not representable")
            if int((self.ref_off.astype(np.int64) + self.ref_len).max()) > self.alleles.size or \
               int((self.alt_off.astype(np.int64) + self.alt_len).max()) > self.alleles.size:
                raise ValueError("allele offsets exceed the pool")

    def slice(self, lo: int, hi: int) -> "VariantTable":
        """Rows [lo, hi) with a re-based private allele pool (used for rank shards)."""
        kw = {c: np.ascontiguousarray(getattr(self, c)[lo:hi]) for c in self.COLS}
        if hi > lo:
            a0 = int(min(kw["ref_off"].min(), kw["alt_off"].min()))
            a1 = int(max((kw["ref_off"].astype(np.int64) + kw["ref_len"]).max(),
                         (kw["alt_off"].astype(np.int64) + kw["alt_len"]).max()))
        else:
            a0 = a1 = 0
        kw["ref_off"] = (kw["ref_off"] - np.uint32(a0)).astype(np.uint32)
        kw["alt_off"] = (kw["alt_off"] - np.uint32(a0)).astype(np.uint32)
        return VariantTable(alleles=np.ascontiguousarray(self.alleles[a0:a1]), **kw)

    def keys(self) -> np.ndarray:
        """Sorted u64 locus keys (contig << 32 | pos), the blacklist key format."""
        return (self.contig.astype(np.uint64) << np.uint64(32)) | self.pos.astype(np.uint64)


@dataclass
class FlatForest:
    """One tree ensemble in the pointer layout the C ABI takes (ugvc_model_upload).

    node i: feature[i] < 0 marks a leaf whose payload row is left[i]; otherwise
    go left when  x[feature] <= threshold  (MODEL_RF)  or  x[feature] < threshold
    (MODEL_GBT), else right.  `threshold` is f32: for sklearn trees it is the largest f32
    that is <= the stored f64 threshold, which decides identically for every f32 input.
    leaf_value rows: MODEL_RF (p_class0, p_class1) as f64; MODEL_GBT (margin, 0)."""
    kind: int
    feature: np.ndarray                 # i32 [n_nodes]
    threshold: np.ndarray               # f32 [n_nodes]
    left: np.ndarray                    # i32 [n_nodes]
    right: np.ndarray                   # i32 [n_nodes]
    tree_root: np.ndarray               # i32 [n_trees]  node index of each root
    leaf_value: np.ndarray              # f64 [n_leaves, 2]
    n_features: int
    base_score: float = 0.0             # GBT: margin offset
    max_depth: int = 0

    @property
    def n_trees(self) -> int:
        return int(self.tree_root.size)


@dataclass
class FilterResult:
    tree_score: np.ndarray              # f32
    filter: np.ndarray                  # u8  FILTER_PASS / FILTER_LOW_SCORE
    flags: np.ndarray                   # u8  FLAG_* | track bits
