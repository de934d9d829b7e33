"""Mint tests/golden/synth_rf_v1.npz: the frozen synthetic filtering model of SURVEY.md §8(d).

Forest T = 40 trees, max depth 8, F = 20 features, one per variant-type group
{snp, h-indel, non-h-indel}, fitted with scikit-learn on a 200 k labelled synthetic sample
(seed 7); plus an XGBoost-shaped additive ensemble (T = 100, depth 6; C5) distilled from
sklearn GradientBoostingClassifier trees.  Features come from the CPU oracle (this is a
fixture generator, not product code).  Labels ~ Bernoulli(sigmoid(w . standardised x)).
Usage: python tests/golden/make_synth_model.py
"""
import os
import sys

import numpy as np
from sklearn.ensemble import GradientBoostingClassifier, RandomForestClassifier

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from variantcalling_amd import model_io, schema as S, synth  # noqa: E402

cs = synth.make_callset(200_000, seed=7)
ft = O.featurize(cs.variants, cs.ref, cs.runs, cs.tracks, "TGCA", 10, 10)
X, group = ft["X"], ft["group"]
rng = np.random.default_rng(7)
w = rng.normal(size=X.shape[1])
Xs = (X - X.mean(0)) / (X.std(0) + 1e-9)
y = (rng.random(X.shape[0]) < 1.0 / (1.0 + np.exp(-(Xs @ w) * 0.8))).astype(np.int64)

rf, gbt = [], []
for g in range(S.N_GROUPS):
    m = group == g
    clf = RandomForestClassifier(n_estimators=40, max_depth=8, random_state=7 + g, n_jobs=1).fit(X[m], y[m])
    f = model_io.flatten_sklearn(clf)
    p0, p1 = O.forest_predict(f, X[m])
    pp = clf.predict_proba(X[m])
    assert np.array_equal(p1, pp[:, 1]) and np.array_equal(p0, pp[:, 0]), "flat forest != sklearn"
    rf.append(f)
    print(S.GROUP_NAMES[g], "rows", int(m.sum()), "nodes", f.feature.size, "leaves", f.leaf_value.shape[0],
          "depth", f.max_depth, "train acc", float((clf.predict(X[m]) == y[m]).mean()))
    # XGBoost-shaped: regression trees of a GBM on the log-odds, leaves pre-scaled by the rate
    gb = GradientBoostingClassifier(n_estimators=100, max_depth=6, learning_rate=0.1, subsample=0.5,
                                    random_state=11 + g).fit(X[m][:30000], y[m][:30000])
    trees = []
    for est in gb.estimators_[:, 0]:
        t = est.tree_
        leaf = t.children_left == -1
        thr = model_io.f32_ceil(np.nextafter(t.threshold, np.inf))   # x <= thr  <=>  x < next(thr)
        trees.append((np.where(leaf, -1, t.feature), np.where(leaf, 0, thr), t.children_left,
                      t.children_right, (t.value[:, 0, 0] * gb.learning_rate).astype(np.float32)))
    prior = float(gb.init_.class_prior_[1])
    gbt.append(model_io.make_gbt(trees, X.shape[1], base_margin=float(np.log(prior / (1 - prior)))))

dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "synth_rf_v1.npz")
model_io.save_models(dst, {"rf_model_ignore_gt_incl_hpol_runs": rf, "xgb_model_ignore_gt_incl_hpol_runs": gbt},
                     meta=dict(seed=7, n=200000, trees=40, depth=8, features=list(S.feature_names(3))))
print(dst, os.path.getsize(dst))
