"""annotate_concordance: the per-variant context columns of the reference's call tables, from the GPU feature matrix.

In the reference `annotate_concordance(df, fasta, runfile=..., flow_order=..., annotate_intervals=...)` (absent
submodule; call shape in SURVEY.md appendix A, its test input is the in-tree
test/resources/unit/comparison/test_vcf_pipeline_utils/annotate_concordance_h5_input.hdf) chains `classify_indel`,
`is_hmer_indel`, `get_motif_around`, `get_gc_content`, `annotate_cycle_skip`, `close_to_hmer_run` / `inside_hmer_run` and
one boolean per interval file over a pandas frame, row by row (/root/reference/ugvc/pipelines/run_no_gt_report.py:90-143
uses the same featurizer and names the produced columns).  Here one launch (`ugvc_feature_matrix`, the K1 kernel with
the quantisation switched off) computes them for the whole table on the MI355X; this module only maps the numeric
feature columns back to the frame vocabulary: `indel_classify` None / "ins" / "del", `hmer_indel_nuc` a base letter,
motifs as 5-base strings, `cycleskip_status` by name (header.txt:3382), interval columns under their file stems."""
from __future__ import annotations

import numpy as np

from . import schema as S
from .io import concordance, h5

_BASES = np.array(["N", "A", "C", "G", "T"], dtype=object)
_CLASSIFY = np.array([None, "ins", "del"], dtype=object)
_CSS = np.array(S.CSS_NAMES, dtype=object)


def motif_strings(codes: np.ndarray) -> np.ndarray:
    """Base-5 motif codes (first base most significant, N = 0) -> MOTIF_SIZE-letter strings."""
    codes = np.asarray(codes).astype(np.int64)
    out = np.full(codes.shape, "", dtype=object)
    for k in range(S.MOTIF_SIZE):
        out = out + _BASES[(codes // 5 ** (S.MOTIF_SIZE - 1 - k)) % 5]
    return out


def columns_from_features(X: np.ndarray, group: np.ndarray, track_names=()) -> h5.Frame:
    """N x F feature matrix (S.feature_names order) -> annotation columns in the reference's vocabulary."""
    names = S.feature_names(len(track_names))
    if X.ndim != 2 or X.shape[1] != len(names):
        raise ValueError(f"feature matrix is {X.shape}, expected N x {len(names)} for {len(track_names)} interval track(s)")
    col = {n: X[:, j] for j, n in enumerate(names)}
    icl = col["indel_classify"].astype(np.int64)
    hnuc = col["hmer_indel_nuc"].astype(np.int64)
    out = h5.Frame()
    out["indel"] = icl != S.INDEL_NONE
    out["indel_classify"] = _CLASSIFY[icl]
    out["indel_length"] = col["indel_length"].astype(np.int64)
    out["hmer_indel_length"] = col["hmer_indel_length"].astype(np.int64)
    out["hmer_indel_nuc"] = np.where(col["hmer_indel_length"] > 0, _BASES[hnuc], None).astype(object)
    out["left_motif"] = motif_strings(col["left_motif"])
    out["right_motif"] = motif_strings(col["right_motif"])
    # the feature is f32(k / window): hand back k / window itself, as the reference's Python division gives it
    out["gc_content"] = np.rint(col["gc_content"].astype(np.float64) * S.GC_WINDOW) / S.GC_WINDOW
    out["cycleskip_status"] = _CSS[col["cycleskip_status"].astype(np.int64)]
    out["inside_hmer_run"] = col["inside_hmer_run"] > 0
    out["close_to_hmer_run"] = col["close_to_hmer_run"] > 0
    out["variant_type"] = np.array(S.GROUP_NAMES, dtype=object)[np.asarray(group).astype(np.int64)]
    for t, name in enumerate(track_names):
        out[name or f"track{t}"] = col[f"track{t}"] > 0
    return out


def annotate_concordance(frame: h5.Frame, engine, ref: S.Reference, runs=None, tracks=(), flow_order: str = "TGCA",
                         hpol_filter_length_dist=(10, 10), is_mutect: bool = False):
    """-> (frame with the annotation columns added, list of the interval column names).  Rows without a call (missed
    truth variants) or off the reference keep None / NaN / False in the new columns, as rows the reference's featurizer
    cannot place do."""
    from .engine import configure
    vt, rows, _ = concordance.frame_to_table(frame, ref.names, is_mutect=is_mutect)
    configure(engine, ref, runs, list(tracks), None, [None] * S.N_GROUPS, flow_order, int(hpol_filter_length_dist[0]),
              int(hpol_filter_length_dist[1]), True)
    X, group = engine.feature_matrix(vt)
    stems = [t.name or f"track{k}" for k, t in enumerate(tracks)]
    ann = columns_from_features(X, group, stems)
    n = frame.n_rows
    out = h5.Frame(frame, index=frame.index, index_names=frame.index_names)
    for name, vals in ann.items():
        if vals.dtype == object:
            full = np.full(n, None, dtype=object)
        elif vals.dtype.kind == "b":
            full = np.zeros(n, bool)
        elif vals.dtype.kind == "f":
            full = np.full(n, np.nan)
        else:
            full = np.zeros(n, vals.dtype)
        full[rows] = vals
        out[name] = full
    return out, stems
