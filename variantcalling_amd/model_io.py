"""Model import: scikit-learn forests / trees and XGBoost-style JSON -> flat node tables.

The reference hands a pickled dict of named models from train_models_pipeline to
filter_variants_pipeline (`--model_file`, `--model_name`: docs/filter_variants_pipeline.md:26-29;
names such as `rf_model_ignore_gt_incl_hpol_runs`: docs/howto-callset-filter.md:114; libs
scikit-learn / xgboost: setup/environment.yml:399,354).  The engine wants plain arrays, so
this module flattens estimators into `schema.FlatForest` (pointer layout) - host logic only.
"""
from __future__ import annotations

import json
import pickle

import numpy as np

from . import schema as S


def f32_floor(thr64: np.ndarray) -> np.ndarray:
    """Largest float32 <= each float64 threshold.

    scikit-learn compares a float32 feature with a float64 threshold (`X[i, f] <= thr`);
    for every float32 x:  x <= thr64  <=>  x <= f32_floor(thr64), so the f32 table decides
    identically (bit-exact FILTER) at half the node size."""
    t = thr64.astype(np.float32)
    over = t.astype(np.float64) > thr64
    t[over] = np.nextafter(t[over], np.float32(-np.inf))
    return t


def f32_ceil(thr64: np.ndarray) -> np.ndarray:
    """Smallest float32 >= each float64 threshold (x < thr64 <=> x < f32_ceil(thr64))."""
    t = thr64.astype(np.float32)
    under = t.astype(np.float64) < thr64
    t[under] = np.nextafter(t[under], np.float32(np.inf))
    return t


def _depth(left, right, feature, root):
    best, stack = 0, [(root, 0)]
    while stack:
        i, d = stack.pop()
        if feature[i] < 0:
            best = max(best, d)
        else:
            stack.append((left[i], d + 1))
            stack.append((right[i], d + 1))
    return best


# names the reference's frames give the engine's features (SURVEY.md appendix A "features fed to the forest";
# ugvc/reports/report_data_loader.py:24-28,67-94): an estimator fitted on a named frame is re-indexed by name
FEATURE_ALIASES = {"ad_0": "ad_ref", "ad[0]": "ad_ref", "ad_1": "ad_alt", "ad[1]": "ad_alt", "af": "vaf", "af_0": "vaf", "af[0]": "vaf",
                   "max_vaf": "vaf", "indel": "indel_classify", "hmer_length": "hmer_indel_length"}


def feature_permutation(model, track_names: list | None = None):
    """Engine feature index of every input column of an estimator that carries `feature_names_in_`, or None when it
    has no names (then its columns are taken to be in the engine's order).  Interval-annotation columns are matched to
    `track_names` (the BED stems of --annotate_intervals, in command-line order)."""
    names = getattr(model, "feature_names_in_", None)
    if names is None:
        return None
    ours = {n: i for i, n in enumerate(S.BASE_FEATURES)}
    tracks = {str(n): S.N_BASE_FEATURES + t for t, n in enumerate(track_names or [])}
    perm, unknown = [], []
    for n in (str(x) for x in names):
        key = FEATURE_ALIASES.get(n.lower(), n.lower())
        if key in ours:
            perm.append(ours[key])
        elif n in tracks:
            perm.append(tracks[n])
        elif key.startswith("track") and key[5:].isdigit():
            perm.append(S.N_BASE_FEATURES + int(key[5:]))
        else:
            unknown.append(n)
    if unknown:
        raise ValueError(f"the model was fitted on columns this engine does not compute: {unknown}; engine features: "
                         f"{list(S.BASE_FEATURES)} + one per --annotate_intervals file {list(track_names or [])}")
    return np.asarray(perm, np.int32)


def flatten_sklearn(model, n_features: int | None = None, track_names: list | None = None) -> S.FlatForest:
    """RandomForestClassifier / ExtraTrees / DecisionTreeClassifier (binary) -> FlatForest."""
    perm = feature_permutation(model, track_names)
    # (a histogram gradient-boosting classifier - the XGBoost-style additive ensemble train_models_pipeline fits as `xgb_model_*`)
    f = flatten_hist_gbt(model, n_features) if hasattr(model, "_predictors") and hasattr(model, "_baseline_prediction") else _flatten_sklearn(model, n_features)
    if perm is not None:
        inner = f.feature >= 0
        f.feature = np.where(inner, perm[np.clip(f.feature, 0, perm.size - 1)], f.feature).astype(np.int32)
        f.n_features = max(int(f.n_features), int(perm.max()) + 1)
    return f


class _TreeArrays:
    """What `_flatten_sklearn` reads of a fitted tree, from the compiled `sklearn.tree._tree.Tree` or from its pickled
    state held as data (legacy_pickle: a <= 1.2 node array that this scikit-learn refuses to load)."""
    def __init__(self, t):
        if hasattr(t, "children_left"):
            self.node_count, self.max_depth = int(t.node_count), int(t.max_depth)
            self.children_left, self.children_right = t.children_left, t.children_right
            self.feature, self.threshold, self.value = t.feature, t.threshold, t.value
            return
        nodes = getattr(t, "nodes", None)
        values = getattr(t, "values", None)
        if nodes is None or values is None or getattr(nodes, "dtype", None) is None or nodes.dtype.names is None:
            raise ValueError("unreadable scikit-learn tree state in the pickle (expected `nodes` / `values` arrays)")
        self.node_count = int(getattr(t, "node_count", nodes.shape[0]))
        nodes = nodes[: self.node_count]
        self.children_left = np.asarray(nodes["left_child"], np.int64)
        self.children_right = np.asarray(nodes["right_child"], np.int64)
        self.feature = np.asarray(nodes["feature"], np.int64)
        self.threshold = np.asarray(nodes["threshold"], np.float64)
        self.value = np.asarray(values, np.float64)[: self.node_count]
        self.max_depth = int(getattr(t, "max_depth", 0))


def tree_arrays(t) -> _TreeArrays:
    return _TreeArrays(t)


def _flatten_sklearn(model, n_features: int | None = None) -> S.FlatForest:
    ests = list(model.estimators_) if hasattr(model, "estimators_") else [model]
    classes = list(getattr(model, "classes_", [0, 1]))
    if len(classes) > 2:
        raise ValueError("only binary (fp/tp) classifiers are supported")
    feats, thrs, lefts, rights, roots, leaves = [], [], [], [], [], []
    node_base = leaf_base = 0
    depth = 0
    for e in ests:
        t = tree_arrays(e.tree_)
        n = t.node_count
        is_leaf = t.children_left == -1
        leaf_id = np.cumsum(is_leaf) - 1 + leaf_base
        feat = np.where(is_leaf, -1, t.feature).astype(np.int32)
        left = np.where(is_leaf, leaf_id, t.children_left + node_base).astype(np.int32)
        right = np.where(is_leaf, 0, t.children_right + node_base).astype(np.int32)
        thr = np.where(is_leaf, 0.0, t.threshold)
        val = np.asarray(t.value[is_leaf, 0, :], dtype=np.float64)
        # scikit-learn <= 1.2 (the reference pins 1.2.2, setup/environment.yml:399) stores weighted sample COUNTS per
        # leaf and `predict_proba` divides every row by its sum (zero sums stay); >= 1.3 stores the fractions and
        # returns them untouched.  Count-valued leaves are recognised by a row sum above 1 and normalised exactly
        # as that predict_proba does, so a reference `--model_file` pickle scores as it does in the reference.
        rs = val.sum(axis=1, keepdims=True)
        if val.size and float(rs.max()) > 1.0 + 1e-9:
            rs[rs == 0.0] = 1.0
            val = val / rs
        if val.shape[1] == 1:                       # single-class tree
            only = int(e.classes_[0]) if hasattr(e, "classes_") else 0
            full = np.zeros((val.shape[0], 2))
            full[:, 1 if only == classes[-1] and len(classes) == 2 else 0] = val[:, 0]
            val = full
        feats.append(feat); thrs.append(f32_floor(thr)); lefts.append(left); rights.append(right)
        roots.append(node_base); leaves.append(val.astype(np.float64))
        depth = max(depth, int(t.max_depth))
        node_base += n
        leaf_base += int(is_leaf.sum())
    nf = n_features or int(getattr(model, "n_features_in_", 0))
    return S.FlatForest(S.MODEL_RF, np.concatenate(feats), np.concatenate(thrs),
                        np.concatenate(lefts), np.concatenate(rights),
                        np.array(roots, dtype=np.int32), np.concatenate(leaves, axis=0),
                        n_features=nf, max_depth=depth)


def flatten_xgb_json(doc: dict | str) -> S.FlatForest:
    """XGBoost `save_model(...json)` document (binary:logistic) -> FlatForest (MODEL_GBT).

    xgboost itself is not installable here; the JSON schema fields used are
    learner.gradient_booster.model.trees[*].{split_indices, split_conditions, left_children,
    right_children, base_weights} and learner.learner_model_param.base_score."""
    if isinstance(doc, str):
        doc = json.loads(doc)
    learner = doc["learner"]
    bs = learner["learner_model_param"]["base_score"]
    if isinstance(bs, str):                                     # "5E-1"; newer writers: "[5E-1]" (one entry per target)
        inner = bs.strip().lstrip("[").rstrip("]").split(",")
        if len(inner) != 1:
            raise ValueError("multi-target XGBoost models are not supported (base_score has several entries)")
        bs = inner[0]
    elif isinstance(bs, (list, tuple)):
        if len(bs) != 1:
            raise ValueError("multi-target XGBoost models are not supported (base_score has several entries)")
        bs = bs[0]
    base = float(bs)
    for tr in learner["gradient_booster"]["model"]["trees"]:
        if tr.get("categories") or any(int(x) != 0 for x in tr.get("split_type", [])):
            raise ValueError("categorical splits are not supported (the hot path's features are numeric)")
    base = min(max(base, 1e-7), 1 - 1e-7)
    margin0 = float(np.log(base / (1.0 - base)))
    feats, thrs, lefts, rights, roots, leaves = [], [], [], [], [], []
    nb = lb = 0
    for tr in learner["gradient_booster"]["model"]["trees"]:
        lc = np.asarray(tr["left_children"], dtype=np.int64)
        rc = np.asarray(tr["right_children"], dtype=np.int64)
        is_leaf = lc == -1
        leaf_id = np.cumsum(is_leaf) - 1 + lb
        feat = np.where(is_leaf, -1, np.asarray(tr["split_indices"])).astype(np.int32)
        cond = np.asarray(tr["split_conditions"], dtype=np.float64)
        thr = np.where(is_leaf, 0.0, cond).astype(np.float32)   # xgboost stores f32 already
        left = np.where(is_leaf, leaf_id, lc + nb).astype(np.int32)
        right = np.where(is_leaf, 0, rc + nb).astype(np.int32)
        val = np.zeros((int(is_leaf.sum()), 2))
        val[:, 0] = cond[is_leaf].astype(np.float32)            # leaf value lives in split_conditions
        feats.append(feat); thrs.append(thr); lefts.append(left); rights.append(right)
        roots.append(nb); leaves.append(val)
        nb += lc.size
        lb += int(is_leaf.sum())
    f = S.FlatForest(S.MODEL_GBT, np.concatenate(feats), np.concatenate(thrs),
                     np.concatenate(lefts), np.concatenate(rights), np.array(roots, dtype=np.int32),
                     np.concatenate(leaves, axis=0),
                     n_features=int(learner["learner_model_param"]["num_feature"]),
                     base_score=margin0)
    f.max_depth = max(_depth(f.left, f.right, f.feature, int(r)) for r in f.tree_root)
    return f


def make_gbt(trees: list, n_features: int, base_margin: float = 0.0) -> S.FlatForest:
    """Build a MODEL_GBT FlatForest from (feature, threshold, left, right, value) array tuples
    (children -1 at leaves) - used for the XGBoost-shaped C5 ensemble."""
    doc_trees = []
    for feat, thr, lc, rc, val in trees:
        cond = np.where(np.asarray(lc) == -1, val, thr)
        doc_trees.append(dict(split_indices=list(map(int, np.maximum(feat, 0))),
                              split_conditions=[float(np.float32(c)) for c in cond],
                              left_children=list(map(int, lc)), right_children=list(map(int, rc))))
    p = 1.0 / (1.0 + np.exp(-base_margin))
    doc = dict(learner=dict(learner_model_param=dict(base_score=str(p), num_feature=str(n_features)),
                            gradient_booster=dict(model=dict(trees=doc_trees))))
    return flatten_xgb_json(doc)


def flatten_hist_gbt(model, n_features: int | None = None) -> S.FlatForest:
    """A fitted scikit-learn `HistGradientBoostingClassifier` (binary) -> FlatForest (MODEL_GBT), the XGBoost-style table the engine
    scores (f32 `x < threshold` goes left, f32 additive margin from `base_score`, sigmoid).

    scikit-learn sends a row left when `x <= num_threshold` (f64 threshold, the f32 feature widened): for every finite f32 x that
    is `x <= t` with t = the largest f32 not above the threshold, i.e. `x < nextafter(t, +inf)` - the threshold stored here, so
    every finite input takes the same path.  NaN inputs go RIGHT here (the engine's compare is false), which is scikit-learn's
    choice too when it saw no missing value in that feature during the fit... only if the right child is the larger one: the
    hot path's features are never NaN after featurisation (QUAL / SOR default to 0), so this is not exercised.  Leaf values carry
    the learning rate already; the margins are added in f32 here and in f64 by scikit-learn: equal to ~1e-6 relative, the class
    (margin > 0) equal except inside that sliver."""
    preds = getattr(model, "_predictors", None)
    if preds is None or any(len(it) != 1 for it in preds):
        raise ValueError("expected a fitted binary HistGradientBoostingClassifier")
    trees = []
    for (p,) in preds:
        nd = p.nodes
        if nd["is_categorical"].any():
            raise ValueError("categorical splits are not supported (the hot path's features are numeric)")
        leaf = nd["is_leaf"].astype(bool)
        t32 = np.nextafter(f32_floor(nd["num_threshold"].astype(np.float64)), np.float32(np.inf)).astype(np.float32)
        feat = np.where(leaf, -1, nd["feature_idx"].astype(np.int64)).astype(np.int32)
        lc = np.where(leaf, -1, nd["left"].astype(np.int64))
        rc = np.where(leaf, -1, nd["right"].astype(np.int64))
        trees.append((feat, np.where(leaf, np.float32(0), t32), lc, rc, nd["value"].astype(np.float64)))
    nf = int(n_features if n_features is not None else model.n_features_in_)
    f = make_gbt(trees, nf, float(np.asarray(model._baseline_prediction).ravel()[0]))
    return f


def make_threshold_model(a, b, label, weight=None, i_a: int = 0, i_b: int = 1, n_features: int = 2, k: int = 16) -> S.FlatForest:
    """The "simple model" of docs/howto-callset-filter.md:129 (`threshold_model_*`, :139): a confidence score from two
    features - QUAL (10 * max TLOD under --is_mutect) and SOR.  BUILDER-DEFINED (the reference's class lives in the
    absent submodule): up to k bins per feature at quantiles of the labelled calls; a cell's score is its smoothed
    true-call fraction (tp + 1) / (n + 2), closed monotonically (the maximum over the cells that are no better on either
    axis), so the score never falls with QUAL nor rises with SOR; a call passes when its score exceeds one half.  Returned as a single-tree forest (search over the QUAL cuts, then over the
    SOR cuts), so it runs on the same kernels as the forests and pickles as plain arrays."""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    y = np.asarray(label).astype(bool)
    w = np.ones(a.size) if weight is None else np.asarray(weight, np.float64)

    def cuts(x):
        q = np.unique(np.quantile(x, np.linspace(0, 1, k + 1)[1:-1], method="inverted_cdf")) if x.size else np.zeros(0)
        return q.astype(np.float32)
    ca, cb = cuts(a), cuts(b)                                      # bin i of x: number of cuts < x  (x <= cut goes left)
    ia, ib = np.searchsorted(ca, a, side="left"), np.searchsorted(cb, b, side="left")
    na, nb = ca.size + 1, cb.size + 1
    tp = np.zeros((na, nb))
    tot = np.zeros((na, nb))
    np.add.at(tp, (ia, ib), w * y)
    np.add.at(tot, (ia, ib), w)
    # smoothed true-call fraction of every cell, then its monotone closure: a call scores at least what any cell that is
    # no better on either axis (lower or equal QUAL bin, higher or equal SOR bin) scores
    score = (tp + 1.0) / (tot + 2.0)
    score = np.maximum.accumulate(score, axis=0)
    score = np.maximum.accumulate(score[:, ::-1], axis=1)[:, ::-1]
    feature, threshold, left, right, leaves = [], [], [], [], []

    def leaf(i, j):
        feature.append(-1); threshold.append(0.0); left.append(len(leaves)); right.append(0)
        leaves.append((1.0 - score[i, j], score[i, j]))
        return len(feature) - 1

    def search(lo, hi, cut, feat, on_bin):
        """bins lo..hi (inclusive) of one axis -> node index; `on_bin(i)` builds what hangs under bin i."""
        if lo == hi:
            return on_bin(lo)
        mid = (lo + hi) // 2                                       # x <= cut[mid]  <=>  bin <= mid
        me = len(feature)
        feature.append(feat); threshold.append(float(cut[mid])); left.append(-1); right.append(-1)
        left[me] = search(lo, mid, cut, feat, on_bin)
        right[me] = search(mid + 1, hi, cut, feat, on_bin)
        return me

    root = search(0, na - 1, ca, i_a, lambda i: search(0, nb - 1, cb, i_b, lambda j: leaf(i, j)))
    depth = int(np.ceil(np.log2(max(na, 1)))) + int(np.ceil(np.log2(max(nb, 1))))
    return S.FlatForest(S.MODEL_RF, np.array(feature, np.int32), np.array(threshold, np.float32), np.array(left, np.int32),
                        np.array(right, np.int32), np.array([root], np.int32), np.array(leaves, np.float64).reshape(-1, 2),
                        n_features=n_features, max_depth=depth)


# ---------------------------------------------------------------- persistence of flat models
def save_models(path: str, models: dict, meta: dict | None = None) -> None:
    """{name: [FlatForest per group]} -> one .npz (the frozen synthetic model of SURVEY.md §8(d))."""
    out = {"__meta__": np.frombuffer(json.dumps(meta or {}).encode(), dtype=np.uint8)}
    names = []
    for name, groups in models.items():
        names.append(name)
        for g, f in enumerate(groups):
            p = f"{name}/{g}/"
            out[p + "hdr"] = np.array([f.kind, f.n_features, f.max_depth], dtype=np.int64)
            out[p + "base"] = np.array([f.base_score], dtype=np.float64)
            for k in ("feature", "threshold", "left", "right", "tree_root", "leaf_value"):
                out[p + k] = getattr(f, k)
    out["__names__"] = np.frombuffer("\n".join(names).encode(), dtype=np.uint8)
    np.savez_compressed(path, **out)


def load_models(path: str) -> dict:
    z = np.load(path)
    names = bytes(z["__names__"]).decode().split("\n")
    models = {}
    for name in names:
        groups = []
        for g in range(S.N_GROUPS):
            p = f"{name}/{g}/"
            if p + "hdr" not in z:
                break
            kind, nf, md = (int(x) for x in z[p + "hdr"])
            groups.append(S.FlatForest(kind, z[p + "feature"], z[p + "threshold"], z[p + "left"],
                                       z[p + "right"], z[p + "tree_root"], z[p + "leaf_value"],
                                       n_features=nf, base_score=float(z[p + "base"][0]), max_depth=md))
        models[name] = groups
    return models


def load_model_file(path: str, model_name: str | None = None, track_names: list | None = None):
    """`--model_file` / `--model_name`: a .npz of flat models, or a pickle holding a dict of
    {name: model}; a model is either [estimator per group] / {group_name: estimator} or one
    estimator used for every group.  `track_names`: the BED stems of --annotate_intervals in command-line order - an
    estimator fitted on a named frame finds its interval-annotation columns (`LCR-hs38`, `exome.twist` ...:
    ugvc/reports/report_data_loader.py:94) by these names."""
    if path.endswith(".npz"):
        models = load_models(path)
    else:
        try:
            with open(path, "rb") as fh:
                raw = pickle.load(fh)
        except (ImportError, AttributeError, ValueError) as exc:
            # a pickle of the reference's own model classes (ugbio_filtering.*, absent here), or of estimators of another
            # scikit-learn generation (ValueError from Tree.__setstate__ - and only that ValueError: a corrupt pickle keeps
            # its own message): read the data without the classes and pull the estimators out of the object graph
            # (legacy_pickle.py)
            from . import legacy_pickle
            if isinstance(exc, ValueError) and not legacy_pickle.is_tree_state_mismatch(exc):
                raise
            raw = legacy_pickle.find_estimators(legacy_pickle.load(path), S.GROUP_NAMES)
            if not raw:
                raise ValueError(f"{path}: no scikit-learn estimator found in the pickle")
        models = {}
        for name, m in (raw.items() if isinstance(raw, dict) else [("model", raw)]):
            if isinstance(m, dict):
                m = [m.get(g) for g in S.GROUP_NAMES]       # a group without a model scores 0 / PASS
            if not isinstance(m, (list, tuple)):
                m = [m] * S.N_GROUPS
            models[name] = [x if x is None or isinstance(x, S.FlatForest) else flatten_sklearn(x, track_names=track_names) for x in m]
    if model_name is None:
        return models
    if model_name not in models:
        raise KeyError(f"model {model_name!r} not in {path}; available: {sorted(models)}")
    return models[model_name]
