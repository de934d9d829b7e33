"""train_models_pipeline: train per-variant-type filtering models; features and evaluation on an MI355X.

Drop-in for `ugbio_filtering.train_models_pipeline.run(argv)` (registered at
/root/reference/ugvc/__main__.py:18,48; script setup.py:41), flags exactly as documented in
docs/train_models_pipeline.md:17-81.  Two modes (docs :5-10): approximate ground truth from a call VCF
(dbSNP id => true positive, `--blacklist` member => false positive) or exact labels.  The N x F feature
matrix is built on the GPU (`ugvc_feature_matrix`); fitting is scikit-learn on the host as in the reference
(random forest / decision tree / the two-feature threshold model / an XGBoost-style gradient-boosted ensemble - `fit_models` -,
one model per variant-type group, exome re-weighting docs :66-72); the
optional `--evaluate_concordance` pass scores on the GPU.
Exact-label input is the comparison HDF5 (per-contig pandas frames with a `classify` column, read by io/h5.py +
io/concordance.py; `--list_of_contigs_to_read` picks the keys) or an `.npz` dump of the SoA table with a `label`
column.  Outputs: PREFIX.pkl (+ PREFIX.npz, the flattened forests), PREFIX.h5 with the keys `training_set` (chrom,
pos, label, group and the F feature columns) and, with --evaluate_concordance, `scored_concordance` and
`optimal_recall_precision`, and PREFIX.stats.csv."""
from __future__ import annotations

import argparse
import csv
import logging
import pickle
import sys

import numpy as np

from .. import evaluate, model_io, schema as S
from ..io import concordance, h5
from ..io import vcf_native as vcfio      # native codec (libugvc_vcf.so); io.vcf is its pure-Python reference
from . import common

logger = logging.getLogger("ugvc")
N_TREES, MAX_DEPTH = 40, 8
# the XGBoost-style additive ensemble (round 6): 100 trees of depth <= 6 - the shape config C5 scores as a leaf-matrix GEMM on MFMA
GBT_TREES, GBT_DEPTH, GBT_RATE = 100, 6, 0.1


def get_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="train_models_pipeline.py", description="Train filtering models on the concordance file")
    ap.add_argument("--input_file", help="Name of the input h5/vcf file. h5 is output of comparison", type=str)
    ap.add_argument("--blacklist", help="blacklist file by which we decide variants as FP", type=str)
    ap.add_argument("--output_file_prefix", help="Output .pkl file with models, .h5 file with results", type=str, required=True)
    ap.add_argument("--mutect", action="store_true")
    ap.add_argument("--evaluate_concordance", help="Should the results of the model be applied to the concordance dataframe",
                    action="store_true")
    ap.add_argument("--apply_model", help="If evaluate_concordance - which model should be applied", type=str)
    ap.add_argument("--evaluate_concordance_contig", help="Which contig the evaluation of the model should be done on", type=str)
    ap.add_argument("--input_interval", help="bed file of intersected intervals from run_comparison pipeline", type=str)
    ap.add_argument("--list_of_contigs_to_read", nargs="*", help="List of contigs to read from the DF", default=[])
    ap.add_argument("--reference", help="Reference genome", type=str, required=True)
    ap.add_argument("--runs_intervals", help="Runs intervals (bed/interval_list)", type=str)
    ap.add_argument("--annotate_intervals", help="interval files for annotation (multiple possible)", type=str,
                    action="append", default=[])
    ap.add_argument("--exome_weight", help="weight of exome variants in comparison to whole genome variant", type=int, default=1)
    ap.add_argument("--flow_order", help="Sequencing flow order (4 cycle)", type=str, default="TGCA")
    ap.add_argument("--exome_weight_annotation", help="annotation name by which we decide the weight of exome variants", type=str)
    ap.add_argument("--vcf_type", help='VCF type - "single_sample" or "joint"', type=str, default="single_sample")
    ap.add_argument("--ignore_filter_status", help="Ignore the `filter` and `tree_score` columns", action="store_true")
    ap.add_argument("--verbosity", help="Verbosity: ERROR, WARNING, INFO, DEBUG", default="INFO")
    ap.add_argument("--device", help="GPU index (MI355X)", type=int, default=0)
    return ap


# FILTER entries that do not mean "a previous filtering round removed this call": evaluate_concordance ignores HPOL_RUN by
# default (ugvc/pipelines/evaluate_concordance.py:44-48); RefCall / missing are the callers' own conventions
_UNFILTERED = frozenset({"", ".", "PASS", "HPOL_RUN", "None", "nan"})


def _was_filtered(strings) -> np.ndarray:
    """True where a FILTER / `filter` entry carries a tag other than PASS / HPOL_RUN (e.g. LOW_SCORE, COHORT_FP)."""
    out = np.zeros(len(strings), bool)
    for i, s in enumerate(strings):
        if s is None or (isinstance(s, float) and s != s):
            continue
        out[i] = any(t not in _UNFILTERED for t in str(s).replace(",", ";").split(";"))
    return out


def _read_labelled(args, ref, bl):
    """-> (VariantTable, label i8: 1 tp / 0 fp / -1 unlabelled).  Calls an earlier filtering round already removed
    (`filter` column / FILTER field with LOW_SCORE, COHORT_FP, ...) are left unlabelled unless --ignore_filter_status
    ("Ignore the `filter` and `tree_score` columns", docs/train_models_pipeline.md:74-76)."""
    if args.input_file.endswith(".npz"):
        z = np.load(args.input_file)
        vt = S.VariantTable(**{c: np.ascontiguousarray(z[c]) for c in S.VariantTable.COLS}, alleles=z["alleles"])
        vt.validate()
        return vt, z["label"].astype(np.int8)
    if args.input_file.endswith((".h5", ".hdf", ".hdf5")):
        fr = concordance.read_concordance(args.input_file, key="all", contigs=args.list_of_contigs_to_read or None)
        vt, rows, label = concordance.frame_to_table(fr, ref.names, is_mutect=args.mutect)
        if not args.ignore_filter_status and "filter" in fr:
            gone = _was_filtered(np.asarray(fr["filter"], dtype=object)[rows])
            logger.info("%d calls carry a FILTER of an earlier round: not used for training (--ignore_filter_status keeps them)", int(gone.sum()))
            label = np.where(gone, -1, label).astype(np.int8)
        return vt, label
    vcf = vcfio.read_vcf(args.input_file, ref.names, is_mutect=args.mutect)
    vt = vcf.table
    if args.vcf_type == "joint":
        vt = _fold_joint_samples(args, ref, vcf)
    label = np.full(vt.n, -1, dtype=np.int8)
    label[vcf.ids] = 1                                    # dbSNP => TP   (docs/train_models_pipeline.md:8-10)
    if bl is not None and bl.size:
        k = vt.keys()
        i = np.minimum(np.searchsorted(bl, k), bl.size - 1)
        label[bl[i] == k] = 0                             # blacklist => FP
    if not args.ignore_filter_status:
        gone = _was_filtered(vcf.orig_filter)
        if gone.any():
            logger.info("%d records carry a FILTER of an earlier round: not used for training (--ignore_filter_status keeps them)", int(gone.sum()))
            label = np.where(gone, -1, label).astype(np.int8)
    return vt, label


def _fold_joint_samples(args, ref, first):
    """`--vcf_type joint` (docs/train_models_pipeline.md:72-73: "VCF type - single_sample or joint"): one feature row per
    RECORD of a multi-sample callset.  How the reference folds the samples is in the absent submodule; BUILDER-DEFINED here,
    in the spirit of io/multiallelic.py (one verdict per record): the cohort's evidence is pooled - AD and DP summed over
    the samples (the pileup a joint caller saw at the site), GQ = the best sample's, GT = the most alternate genotype any
    sample carries; QUAL, INFO/SOR and the alleles are site-level already.  Reads the FORMAT columns of every sample with
    the native codec (one pass per sample)."""
    n_samples = _n_samples(args.input_file)
    vt = first.table
    if n_samples <= 1:
        return vt
    dp, adr, ada = vt.dp.astype(np.int64), vt.ad_ref.astype(np.int64), vt.ad_alt.astype(np.int64)
    gq, gt = vt.gq.copy(), vt.gt.copy()
    for s in range(1, n_samples):
        t = vcfio.read_vcf(args.input_file, ref.names, is_mutect=args.mutect, sample=s).table
        dp += t.dp
        adr += t.ad_ref
        ada += t.ad_alt
        gq = np.maximum(gq, t.gq)
        gt = np.maximum(gt, t.gt)
    i32 = np.iinfo(np.int32).max
    kw = {c: getattr(vt, c) for c in S.VariantTable.COLS}
    kw.update(dp=np.minimum(dp, i32).astype(np.int32), ad_ref=np.minimum(adr, i32).astype(np.int32),
              ad_alt=np.minimum(ada, i32).astype(np.int32), gq=gq, gt=gt)
    logger.info("--vcf_type joint: %d samples pooled per record", n_samples)
    return S.VariantTable(alleles=vt.alleles, **kw)


def _n_samples(path: str) -> int:
    import gzip
    with open(path, "rb") as fh:
        magic = fh.read(2)
    with (gzip.open(path, "rt") if magic == b"\x1f\x8b" else open(path, "rt")) as fh:
        for line in fh:
            if line.startswith("#CHROM"):
                return max(len(line.rstrip("\r\n").split("\t")) - 9, 0)
            if not line.startswith("#"):
                break
    return 0


def _inside_intervals(track: S.IntervalTrack, contig: np.ndarray, pos: np.ndarray) -> np.ndarray:
    """start < pos <= end of some interval of the row's contig (BED coordinates against the 1-based POS, as the
    annotation tracks are read); intervals are merged, so the last start below pos decides."""
    out = np.zeros(pos.size, bool)
    for c in np.unique(contig):
        m = np.flatnonzero(contig == c)
        a, b = int(track.contig_ptr[c]), int(track.contig_ptr[c + 1])
        if b == a:
            continue
        s = np.searchsorted(track.starts[a:b], pos[m], side="left") - 1
        ok = s >= 0
        out[m[ok]] = track.ends[a:b][s[ok]] >= pos[m[ok]]
    return out


def fit_models(X, group, label, weights, hpol_flag):
    """One model per variant-type group and kind.  `xgb_model_*` (round 6): the newer reference tool is XGBoost-based
    (`setup/environment.yml:354`); xgboost itself cannot be installed here, so the additive ensemble is fitted by
    scikit-learn's histogram gradient boosting - the algorithm family of XGBoost's `hist` tree method (binned features, second-order
    logistic loss, shrinkage) - and stored in XGBoost's FORMAT and scoring semantics (f32 `x < threshold`, f32 additive margin,
    sigmoid: model_io.flatten_hist_gbt), which is what the engine scores (v3 traversal, or the leaf-matrix GEMM of config C5).  The
    trees are not the ones xgboost would grow on the same data: BUILDER-DEFINED fit, reference-defined format."""
    from sklearn.ensemble import HistGradientBoostingClassifier, RandomForestClassifier
    from sklearn.tree import DecisionTreeClassifier
    models = {}
    for incl in (True, False):
        use = (label >= 0) & (incl | ~hpol_flag)
        suffix = "ignore_gt_" + ("incl" if incl else "excl") + "_hpol_runs"
        rf, dt, thr, gb = {}, {}, {}, {}
        i_qual, i_sor = S.BASE_FEATURES.index("qual"), S.BASE_FEATURES.index("sor")
        for g, gname in enumerate(S.GROUP_NAMES):
            m = use & (group == g)
            if m.sum() < 2 or np.unique(label[m]).size < 2:
                logger.warning("group %s (%s): not enough labelled variants of both classes, no model", gname, suffix)
                continue
            rf[gname] = RandomForestClassifier(n_estimators=N_TREES, max_depth=MAX_DEPTH, random_state=g, n_jobs=-1).fit(
                X[m], label[m], sample_weight=weights[m])
            dt[gname] = DecisionTreeClassifier(max_depth=MAX_DEPTH, random_state=g).fit(X[m], label[m], sample_weight=weights[m])
            gb[gname] = HistGradientBoostingClassifier(max_iter=GBT_TREES, max_depth=GBT_DEPTH, max_leaf_nodes=1 << GBT_DEPTH, learning_rate=GBT_RATE,
                                                       early_stopping=False, random_state=g).fit(X[m], label[m], sample_weight=weights[m])
            # the two-feature "simple model" (docs/howto-callset-filter.md:129,139): QUAL (= 10 * TLOD with --mutect) and SOR
            thr[gname] = model_io.make_threshold_model(X[m, i_qual], X[m, i_sor], label[m], weights[m], i_qual, i_sor, X.shape[1])
        models["rf_model_" + suffix] = rf
        models["dt_model_" + suffix] = dt
        models["threshold_model_" + suffix] = thr
        models["xgb_model_" + suffix] = gb
    return models


def _flatten(model: dict):
    return [(model[g] if isinstance(model[g], S.FlatForest) else model_io.flatten_sklearn(model[g])) if g in model else None
            for g in S.GROUP_NAMES]


def run(argv: list[str]):
    """Train filtering models on the concordance file"""
    args = get_parser().parse_args(argv[1:])
    logger.setLevel(getattr(logging, str(args.verbosity).upper(), logging.INFO))
    if not args.input_file:
        raise ValueError("--input_file is required")
    from ..engine import Engine, configure     # fails loudly if the library or the GPU is missing

    if args.vcf_type not in ("single_sample", "joint"):
        raise ValueError(f'--vcf_type {args.vcf_type!r}: "single_sample" or "joint" (docs/train_models_pipeline.md:72-73)')
    if args.vcf_type == "joint" and args.input_file.endswith((".h5", ".hdf", ".hdf5", ".npz")):
        logger.info("--vcf_type joint: the input is a labelled table (one row per record already); nothing to fold")
    ref, runs, tracks, bl = common.load_side_tables(args.reference, args.runs_intervals, args.annotate_intervals, args.blacklist)
    vt, label = _read_labelled(args, ref, bl)
    if args.input_interval:
        # "bed file of intersected intervals from run_comparison pipeline" (docs :55-57): only calls inside the compared
        # (high-confidence) region carry a trustworthy label
        from ..io import vcf_native
        region = vcf_native.read_intervals(args.input_interval, ref.names, merge=True)
        inside = _inside_intervals(region, vt.contig, vt.pos)
        logger.info("--input_interval: %d of %d calls inside the intervals", int(inside.sum()), vt.n)
        label = np.where(inside, label, -1).astype(np.int8)
    if args.list_of_contigs_to_read:
        keep = np.isin(vt.contig, [ref.names.index(c) for c in args.list_of_contigs_to_read if c in ref.names])
        rows = np.flatnonzero(keep)
        if rows.size and rows.size < vt.n:
            parts = [vt.slice(int(a), int(b) + 1) for a, b in zip(rows[np.r_[True, np.diff(rows) > 1]],
                                                                  rows[np.r_[np.diff(rows) > 1, True]])]
            vt = parts[0] if len(parts) == 1 else _concat(parts)
            label = label[rows]
    with Engine(args.device) as eng:
        configure(eng, ref, runs, tracks, None, [None] * S.N_GROUPS, args.flow_order, 10, 10, True)
        X, group = eng.feature_matrix(vt)                  # N x F on the GPU
        names = S.feature_names(len(tracks))
        weights = np.ones(vt.n)
        if args.exome_weight != 1 and args.exome_weight_annotation:
            stems = [t.name for t in tracks]
            if args.exome_weight_annotation not in stems:
                raise ValueError(f"--exome_weight_annotation {args.exome_weight_annotation!r} is not one of {stems}")
            weights[X[:, S.N_BASE_FEATURES + stems.index(args.exome_weight_annotation)] > 0] = args.exome_weight
        hpol = (X[:, names.index("inside_hmer_run")] > 0) | (X[:, names.index("close_to_hmer_run")] > 0)
        logger.info("fitting on %d labelled of %d variants", int((label >= 0).sum()), vt.n)
        models = fit_models(X, group, label, weights, hpol)
        with open(args.output_file_prefix + ".pkl", "wb") as fh:
            pickle.dump(models, fh)
        flat = {k: _flatten(v) for k, v in models.items() if len(v) == S.N_GROUPS}
        if flat:
            model_io.save_models(args.output_file_prefix + ".npz", flat, meta=dict(features=list(names)))
        chrom = np.array(ref.names, dtype=object)[vt.contig]
        results = {"training_set": h5.Frame([("chrom", chrom), ("pos", vt.pos.astype(np.int64)), ("label", label.astype(np.int64)),
                                             ("group", group.astype(np.int64))] + [(nm, X[:, j]) for j, nm in enumerate(names)])}
        if args.evaluate_concordance:
            name = args.apply_model or "rf_model_ignore_gt_incl_hpol_runs"
            if name not in models:
                raise KeyError(f"--apply_model {name!r}; trained: {sorted(models)}")
            flat_applied = _flatten(models[name])
            eng.set_models(flat_applied)
            eng.set_blacklist(bl)
            res = eng.filter_variants(vt)
            # config C5: an additive ensemble of depth <= 6 (the `xgb_model_*` this tool fits) is ALSO evaluated as a leaf-matrix GEMM on
            # the matrix cores over the feature matrix still resident from the fit (`ugvc_forest_gemm3`: one launch for the three
            # variant-type groups) - the margins must decide every FILTER exactly as the scoring pass did and give its TREE_SCORE
            if all(f is None or (f.kind == S.MODEL_GBT and f.max_depth <= 6) for f in flat_applied) and any(f is not None for f in flat_applied):
                rows_g = [np.flatnonzero((group == g)).astype(np.int32) if flat_applied[g] is not None else None for g in range(S.N_GROUPS)]
                margin, ms = eng.forest_gemm3(rows_g)
                has = np.isin(group, [g for g in range(S.N_GROUPS) if flat_applied[g] is not None])
                p_gemm = (1.0 / (1.0 + np.exp(-margin[has].astype(np.float64)))).astype(np.float32)
                if not np.array_equal(margin[has] > 0, res.filter[has] == S.FILTER_PASS) or np.abs(p_gemm - res.tree_score[has]).max() > 2e-6:
                    raise RuntimeError("internal: the leaf-matrix GEMM and the scoring pass disagree on the applied ensemble")
                logger.info("leaf-matrix GEMM (MFMA) over %d rows in %.2f ms: FILTER and TREE_SCORE equal the scoring pass", int(has.sum()), ms)
            sel = label >= 0
            if args.evaluate_concordance_contig and args.evaluate_concordance_contig in ref.names:
                sel &= vt.contig == ref.names.index(args.evaluate_concordance_contig)
            # the accuracy table's counts come from the FILTER column still resident on the GPU (ugvc_eval_counts)
            lab = np.where(sel, label, -1).astype(np.int8)
            bits = evaluate.category_bits(group != S.GROUP_SNP, X[:, names.index("hmer_indel_length")])
            rows = evaluate.accuracy_rows(eng.eval_counts(lab, bits))
            with open(args.output_file_prefix + ".stats.csv", "w", newline="") as fh:
                w = csv.DictWriter(fh, fieldnames=list(rows[0]), delimiter=";")
                w.writeheader()
                w.writerows(rows)
            results["scored_concordance"] = concordance.table_to_frame(vt, ref.names, label, res)
            results["optimal_recall_precision"] = h5.Frame(
                [(k, np.array([r[k] for r in rows], dtype=object if k == "group" else None)) for k in rows[0]])
        h5.write_hdf(args.output_file_prefix + ".h5", results)
    return 0


def _concat(parts):
    kw = {c: np.concatenate([getattr(p, c) for p in parts]) for c in S.VariantTable.COLS}
    base = np.cumsum([0] + [p.alleles.size for p in parts[:-1]])
    kw["ref_off"] = np.concatenate([p.ref_off + np.uint32(b) for p, b in zip(parts, base)]).astype(np.uint32)
    kw["alt_off"] = np.concatenate([p.alt_off + np.uint32(b) for p, b in zip(parts, base)]).astype(np.uint32)
    return S.VariantTable(alleles=np.concatenate([p.alleles for p in parts]), **kw)


if __name__ == "__main__":
    run(sys.argv)
