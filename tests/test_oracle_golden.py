"""CPU tests that pin the ORACLE (the checker) before it is trusted on the GPU box.

Pins available for this path:
  * the reference's own known-answer tests for the SEC statistic and phred helpers
    (test/unit/utils/test_stats_utils.py:18-110, test_math_utils.py:10-23), replayed verbatim;
  * outputs of the reference's in-tree code RUN in the build container
    (tests/golden/reference_run_v1.npz, minted by tests/golden/make_reference_goldens.py):
    calibrate_bridging_snvs.run, multinomial_likelihood_ratio, scale_contingency_table;
  * the real hg38 slice (MD5 == the `M5` tags of the reference's .dict files);
  * scikit-learn `predict_proba` for tree-ensemble scoring (bit-exact);
  * a second, independent restatement in the reference's pandas idiom (oracle/idiom.py);
  * the committed golden outputs of the featurize->score->FILTER oracle (drift guard).
featurize -> score -> FILTER itself stays "parity unpinned" (source + fixtures absent).
"""
import hashlib
import os

import numpy as np
import pytest

import edge_cases as E
from conftest import GOLDEN, real_chr1_reference
from oracle import bridging as B
from oracle import idiom, oracle as O, stats as st
from variantcalling_amd import model_io, schema as S, synth

RF = "rf_model_ignore_gt_incl_hpol_runs"
XGB = "xgb_model_ignore_gt_incl_hpol_runs"


@pytest.fixture(scope="module")
def refrun():
    return np.load(os.path.join(GOLDEN, "reference_run_v1.npz"))


# ------------------------------------------------------------------ reference KATs (verbatim values)
def test_scale_contingency_table_kats():
    for n, exp in ((2, 1), (3, 1), (4, 1), (5, 2), (6, 2), (7, 2), (9, 3)):
        assert st.scale_contingency_table([1, 1, 1], n) == [exp] * 3
        assert st.scale_contingency_table([10, 10, 10], n) == [exp] * 3
    assert st.scale_contingency_table([10, 20, 25], 100) == [18, 36, 45]
    assert st.scale_contingency_table([10, 20, 25], 10) == [2, 4, 5]


def test_correct_multinomial_frequencies_kats():
    assert np.array_equal(np.array([1, 1, 1]) / 3, st.correct_multinomial_frequencies([10, 10, 10]))
    assert np.array_equal(np.array([11, 11, 1]) / 23, st.correct_multinomial_frequencies([10, 10, 0]))


KAT_LIK = [(([4, 4, 4], [4, 4, 4]), 0.0652, 3), (([4, 4, 4], [40, 40, 40]), 0.0652, 3),
           (([40, 40, 40], [40, 40, 40]), 0.0068, 3), (([4, 4, 40], [4, 4, 4]), 3.3e-13, 10),
           (([4, 4, 40], [40, 40, 40]), 3.3e-13, 10), (([10, 10, 10], [1, 10, 40]), 2.1e-10, 10),
           (([40, 10, 1], [1, 10, 40]), 2.7e-53, 40), (([1, 10, 40], [1, 10, 40]), 0.039, 3),
           (([4, 4, 4], [4, 4, 0]), 0.0043, 3), (([4, 4, 40], [0, 0, 0]), 3.3e-13, 3)]
KAT_RATIO = [(([4, 4, 4], [4, 4, 4]), 1, 3), (([4, 4, 4], [40, 40, 40]), 1, 3), (([40, 40, 40], [40, 40, 40]), 1, 3),
             (([4, 4, 40], [4, 4, 4]), 3.3e-13, 10), (([4, 4, 40], [40, 40, 40]), 3.3e-13, 10),
             (([10, 10, 10], [1, 10, 40]), 7.8e-9, 10), (([40, 10, 1], [1, 10, 40]), 6.9e-52, 40),
             (([1, 10, 40], [1, 10, 40]), 1, 3), (([4, 4, 4], [4, 4, 0]), 0.0661, 3),
             (([4, 4, 40], [0, 0, 0]), 9.1e-12, 10)]


def test_multinomial_likelihood_kats():
    for (a, e), val, places in KAT_LIK:
        assert round(abs(st.multinomial_likelihood(a, e) - val), places) == 0, (a, e)
    for (a, e), val, places in KAT_RATIO:
        assert round(abs(st.multinomial_likelihood_ratio(a, e)[1] - val), places) == 0, (a, e)


def test_math_utils_kats():
    assert np.all(st.phred((0.1, 0.01, 0.001)) == np.array([10.0, 20.0, 30.0]))
    assert st.phred_str([0.1, 0.01, 0.001]) == "+5?"
    assert np.allclose(st.unphred((10, 20, 30)), np.array([0.1, 0.01, 0.001]))
    assert np.allclose(st.unphred_str("+5?"), np.array([0.1, 0.01, 0.001]))


# ------------------------------------------------------------------ outputs of the reference run here
def test_sec_statistic_vs_reference_run(refrun):
    lik, ratio = st.sec_batch(refrun["sec_actual"], refrun["sec_expected"])
    assert np.allclose(lik, refrun["sec_lik"], rtol=1e-11, atol=0)
    assert np.allclose(ratio, refrun["sec_ratio"], rtol=1e-11, atol=0)
    for i in (0, 30, 60, 100, 1999):
        l1, r1 = st.multinomial_likelihood_ratio(list(refrun["sec_actual"][i]), list(refrun["sec_expected"][i]))
        assert np.isclose(l1, refrun["sec_lik"][i], rtol=1e-12) and np.isclose(r1, refrun["sec_ratio"][i], rtol=1e-12)
    for t, n, exp in zip(refrun["scale_tables"], refrun["scale_n"], refrun["scale_out"]):
        t = [int(x) for x in t if x >= 0]
        assert [int(x) for x in st.scale_contingency_table(t, int(n))] == [int(x) for x in exp if x >= 0]
    assert np.array_equal(st.phred(refrun["phred_in"]), refrun["phred_out"])
    assert np.array_equal(st.unphred(list(refrun["unphred_in"])), refrun["unphred_out"])


def test_bridging_oracle_vs_reference_run(refrun):
    """oracle/bridging.py against calibrate_bridging_snvs.run executed on stand-in pysam objects."""
    ref = real_chr1_reference()
    vt = E.edge_table(ref, seed=9, n_random=6000)
    use = refrun["bridging_usable"]
    assert use.size == vt.n and use.sum() > 5000
    for h, edge in refrun["bridging_combos"]:
        hm, unf = B.calibrate(vt, ref, refrun["bridging_is_pass"], refrun["bridging_ad_alt"], refrun["bridging_bg_ad"],
                              refrun["bridging_bg_dp"], min_query_hmer_size=int(h), min_distance_from_edge=int(edge))
        assert np.array_equal(hm[use], refrun[f"bridging_hm_{h}_{edge}"][use]), (h, edge)
        assert np.array_equal(unf[use], refrun[f"bridging_unfiltered_{h}_{edge}"][use]), (h, edge)
    assert refrun["bridging_unfiltered_2_0"].sum() > 20


# ------------------------------------------------------------------ real sequence fixture
def test_hg38_fixture_md5():
    z = np.load(os.path.join(GOLDEN, "hg38_chr1_head.npz"))
    table = np.frombuffer(b"NACGT", dtype=np.uint8)
    for name, md5 in (("chr1", "f00eb808cff4be46d8c69c7209038873"), ("chr20", "042e5a811f0a907d7e9f63e558c70f75")):
        codes = synth.load_hg38_slice(name)
        assert hashlib.md5(table[codes].tobytes()).hexdigest() == md5 == bytes(z[f"{name}_md5"]).decode()
    c = synth.load_hg38_slice("chr1")
    # SURVEY.md App. D composition of chr1:1-5,000,000
    assert [int((c == k).sum()) for k in range(5)] == [203509, 1127141, 1280004, 1278358, 1110988]


def test_flow_key_property():
    """test/system/test_collect_hpol_table.py:32-36: cumsum(key) maps flows to bases; the key
    re-expands to the sequence under the cyclic flow order."""
    rng = np.random.default_rng(0)
    for flow in ("TGCA", "ACGT"):
        fo = S.encode_bases(flow)
        for _ in range(200):
            seq = rng.integers(1, 5, size=int(rng.integers(1, 40))).astype(np.uint8)
            key = O.flow_key(seq, fo)
            assert key.sum() == seq.size
            rebuilt = np.concatenate([np.full(h, fo[s % 4], np.uint8) for s, h in enumerate(key)])
            assert np.array_equal(rebuilt, seq)
            assert key[-1] > 0
        assert O.flow_key(np.array([1, 0, 2], np.uint8), fo) is None
    # batch form == scalar form
    fo = S.encode_bases("TGCA")
    seqs = rng.integers(1, 5, size=(300, 12)).astype(np.uint8)
    lens = rng.integers(1, 13, size=300)
    keys, klen = O._flow_keys_batch(seqs, lens, fo)
    for i in range(300):
        k = O.flow_key(seqs[i, : lens[i]], fo)
        assert klen[i] == k.size and np.array_equal(keys[i, : k.size], k)


# ------------------------------------------------------------------ tree ensembles vs scikit-learn
def test_forest_predict_matches_sklearn():
    from sklearn.ensemble import ExtraTreesClassifier, RandomForestClassifier
    from sklearn.tree import DecisionTreeClassifier
    rng = np.random.default_rng(5)
    X = rng.normal(size=(4000, 9)).astype(np.float32)
    X[:, 3] = rng.integers(0, 4, 4000)
    y = (X[:, 0] + 0.5 * X[:, 3] - X[:, 5] ** 2 + rng.normal(0, 0.5, 4000) > 0).astype(int)
    Xt = rng.normal(size=(3000, 9)).astype(np.float32)
    Xt[:, 3] = rng.integers(0, 4, 3000)
    Xt[:40] = X[:40]                                   # values that sit exactly on thresholds' side
    for clf in (RandomForestClassifier(n_estimators=25, max_depth=7, random_state=1),
                ExtraTreesClassifier(n_estimators=10, max_depth=5, random_state=2),
                DecisionTreeClassifier(max_depth=6, random_state=3)):
        clf.fit(X, y)
        f = model_io.flatten_sklearn(clf)
        p0, p1 = O.forest_predict(f, Xt)
        pp = clf.predict_proba(Xt)
        assert np.array_equal(p0, pp[:, 0]) and np.array_equal(p1, pp[:, 1])
        ts, flt = O.score([f, None, None], Xt, np.zeros(Xt.shape[0], np.int64))
        assert np.array_equal(flt == S.FILTER_PASS, clf.predict(Xt) == 1)
        assert np.array_equal(ts, pp[:, 1].astype(np.float32))
        # every threshold itself as an input: x <= thr must agree in f32
        thr = f.threshold[f.feature >= 0]
        Xe = np.zeros((thr.size, 9), np.float32)
        Xe[np.arange(thr.size), f.feature[f.feature >= 0]] = thr
        assert np.array_equal(O.forest_predict(f, Xe)[1], clf.predict_proba(Xe)[:, 1])


def test_count_valued_leaves_score_like_sklearn_1_2_predict_proba():
    """The reference pins scikit-learn 1.2.2 (setup/environment.yml:399): tree_.value holds weighted sample
    counts and predict_proba normalises every leaf row.  Emulated on trees fitted here: the same trees with
    value := fraction x weighted_n_node_samples must flatten to the normalised rows and score as the
    count-normalising predict_proba does."""
    from types import SimpleNamespace
    from sklearn.ensemble import RandomForestClassifier
    rng = np.random.default_rng(9)
    X = rng.normal(size=(3000, 6)).astype(np.float32)
    y = (X[:, 0] - X[:, 2] + rng.normal(0, 0.7, 3000) > 0).astype(int)
    clf = RandomForestClassifier(n_estimators=12, max_depth=6, random_state=4).fit(X, y)
    ests = []
    for e in clf.estimators_:
        t = e.tree_
        counts = t.value * t.weighted_n_node_samples[:, None, None]          # what 1.2.2 stores
        ests.append(SimpleNamespace(classes_=e.classes_, tree_=SimpleNamespace(
            node_count=t.node_count, children_left=t.children_left, children_right=t.children_right, feature=t.feature,
            threshold=t.threshold, value=counts, max_depth=t.max_depth)))
    old = SimpleNamespace(estimators_=ests, classes_=clf.classes_, n_features_in_=6)
    f = model_io.flatten_sklearn(old)
    assert np.allclose(f.leaf_value.sum(axis=1), 1.0, atol=1e-12)
    Xt = rng.normal(size=(2000, 6)).astype(np.float32)
    # predict_proba of 1.2.2: per tree value[leaf] / value[leaf].sum(), averaged over the trees in order
    acc = np.zeros((Xt.shape[0], 2))
    for e, o in zip(clf.estimators_, ests):
        leaf = e.apply(Xt)
        v = o.tree_.value[leaf, 0, :]
        nz = v.sum(axis=1, keepdims=True)
        nz[nz == 0.0] = 1.0
        acc += v / nz
    acc /= len(ests)
    p0, p1 = O.forest_predict(f, Xt)
    assert np.array_equal(p0, acc[:, 0]) and np.array_equal(p1, acc[:, 1])
    # and the fraction-valued (>= 1.3) trees are left untouched
    f_new = model_io.flatten_sklearn(clf)
    assert np.array_equal(O.forest_predict(f_new, Xt)[1], clf.predict_proba(Xt)[:, 1])


def test_gbt_semantics():
    """XGBoost rules: x < thr goes left, f32 margin accumulated in tree order, sigmoid in f32."""
    t0 = (np.array([0, -1, -1]), np.array([0.5, 0, 0], np.float32), np.array([1, -1, -1]), np.array([2, -1, -1]),
          np.array([0, -0.25, 0.75], np.float32))
    t1 = (np.array([1, -1, 0, -1, -1]), np.array([2.0, 0, -1.0, 0, 0], np.float32), np.array([1, -1, 3, -1, -1]),
          np.array([2, -1, 4, -1, -1]), np.array([0, 0.125, 0, -0.5, 0.3], np.float32))
    f = model_io.make_gbt([t0, t1], 2, base_margin=0.1)
    X = np.array([[0.5, 1.0], [0.49999997, 1.0], [0.7, 2.0], [-3.0, 5.0], [np.nan, 0.0]], np.float32)
    m, s = O.forest_predict(f, X)
    b = np.float32(f.base_score)
    exp = np.array([b + np.float32(0.75) + np.float32(0.125), b + np.float32(-0.25) + np.float32(0.125),
                    b + np.float32(0.75) + np.float32(0.3), b + np.float32(-0.25) + np.float32(-0.5),
                    b + np.float32(0.75) + np.float32(0.125)], np.float32)
    assert np.array_equal(m, exp)
    assert np.allclose(s, 1 / (1 + np.exp(-exp.astype(np.float64))), atol=1e-7)


def test_xgboost_json_flattening():
    """`flatten_xgb_json` on a hand-written document in XGBoost's save_model JSON schema (xgboost itself is not
    installable here): split_indices / split_conditions / left_children / right_children per tree, leaf values in
    split_conditions, base_score as a probability -> margin offset logit(base_score)."""
    doc = {"learner": {"learner_model_param": {"base_score": "0.25", "num_feature": "3"},
                       "gradient_booster": {"model": {"trees": [
                           {"left_children": [1, -1, -1], "right_children": [2, -1, -1], "split_indices": [2, 0, 0],
                            "split_conditions": [1.5, -0.4, 0.6], "base_weights": [0, 0, 0]},
                           {"left_children": [1, 3, -1, -1, -1], "right_children": [2, 4, -1, -1, -1],
                            "split_indices": [0, 1, 0, 0, 0], "split_conditions": [10.0, 0.5, 0.2, -0.1, 0.3],
                            "base_weights": [0, 0, 0, 0, 0]}]}}}}
    import json
    for d in (doc, json.dumps(doc)):
        f = model_io.flatten_xgb_json(d)
        assert f.kind == S.MODEL_GBT and f.n_trees == 2 and f.n_features == 3 and f.max_depth == 2
        assert abs(f.base_score - np.log(0.25 / 0.75)) < 1e-12
        assert f.tree_root.tolist() == [0, 3] and f.feature.tolist() == [2, -1, -1, 0, 1, -1, -1, -1]
        X = np.array([[5.0, 0.4, 1.0], [5.0, 0.6, 2.0], [10.0, 0.0, 1.5], [np.nan, 0.0, 0.0]], np.float32)
        m, s = O.forest_predict(f, X)
        b = np.float32(f.base_score)
        # tree 0: x2 < 1.5 ? -0.4 : 0.6 ; tree 1: x0 < 10 ? (x1 < 0.5 ? -0.1 : 0.3) : 0.2 ; NaN never goes left
        exp = np.array([b + np.float32(-0.4) + np.float32(-0.1), b + np.float32(0.6) + np.float32(0.3),
                        b + np.float32(0.6) + np.float32(0.2), b + np.float32(-0.4) + np.float32(0.2)], np.float32)
        assert np.array_equal(m, exp)
    # newer writers store base_score as a one-entry vector; categorical splits and multi-target models are refused
    import copy
    d2 = copy.deepcopy(doc)
    d2["learner"]["learner_model_param"]["base_score"] = "[2.5E-1]"
    assert abs(model_io.flatten_xgb_json(d2).base_score - np.log(0.25 / 0.75)) < 1e-12
    d2["learner"]["learner_model_param"]["base_score"] = "[2.5E-1,5E-1]"
    with pytest.raises(ValueError, match="multi-target"):
        model_io.flatten_xgb_json(d2)
    d3 = copy.deepcopy(doc)
    d3["learner"]["gradient_booster"]["model"]["trees"][0]["split_type"] = [1, 0, 0]
    with pytest.raises(ValueError, match="categorical"):
        model_io.flatten_xgb_json(d3)


# ------------------------------------------------------------------ two restatements agree
@pytest.mark.parametrize("flow", ["TGCA", "GTAC"])
def test_vectorised_oracle_matches_reference_idiom(frozen_models, flow):
    cs = synth.make_callset(2500, genome_len=3_000_000, n_contigs=3, seed=21)
    forests = frozen_models[RF]
    exp = O.filter_variants(cs.variants, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests, flow_order=flow)
    got, df, X = idiom.filter_variants_idiom(cs.variants, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests,
                                             flow_order=flow)
    ft = O.featurize(cs.variants, cs.ref, cs.runs, cs.tracks, flow)
    for j, name in enumerate(S.feature_names(3)):
        assert np.array_equal(X[:, j], ft["X"][:, j]), name
    assert np.array_equal(got.filter, exp.filter) and np.array_equal(got.flags, exp.flags)
    assert np.array_equal(got.tree_score, exp.tree_score)
    assert set(df["cycleskip_status"]) <= set(S.CSS_NAMES)


def test_idiom_matches_on_real_hg38_edges(frozen_models):
    ref = real_chr1_reference()
    vt = E.edge_table(ref, seed=3, n_random=600)
    runs, tracks = E.simple_tracks(ref)
    bl = vt.keys()[::5]
    exp = O.filter_variants(vt, ref, runs, tracks, bl, frozen_models[RF], hpol_len=8, hpol_dist=12)
    got, _, X = idiom.filter_variants_idiom(vt, ref, runs, tracks, bl, frozen_models[RF], hpol_len=8, hpol_dist=12)
    assert np.array_equal(X, O.featurize(vt, ref, runs, tracks, "TGCA", 8, 12)["X"])
    assert np.array_equal(got.filter, exp.filter) and np.array_equal(got.flags, exp.flags)
    assert np.array_equal(got.tree_score, exp.tree_score)


# ------------------------------------------------------------------ drift guard on the committed golden outputs
def test_oracle_reproduces_committed_golden(frozen_models):
    z = np.load(os.path.join(GOLDEN, "filter_golden_v1.npz"))
    ref = real_chr1_reference()
    vt = E.edge_table(ref)
    runs, tracks = E.simple_tracks(ref)
    bl = np.unique(np.concatenate([vt.keys()[::7], vt.keys()[::11] + np.uint64(1)]))
    assert np.array_equal(vt.pos, z["pos"]) and np.array_equal(vt.contig, z["contig"])
    res = O.filter_variants(vt, ref, runs, tracks, bl, frozen_models[RF], hpol_len=8, hpol_dist=12)
    assert np.array_equal(res.filter, z["filter"]) and np.array_equal(res.flags, z["flags"])
    assert np.array_equal(res.tree_score, z["tree_score"])
    assert np.array_equal(O.featurize(vt, ref, runs, tracks, "TGCA", 8, 12)["X"], z["X"])
    # the table exercises every class of edge: all groups, all cycle-skip states, every flag
    ft = O.featurize(vt, ref, runs, tracks, "TGCA", 8, 12)
    assert set(np.unique(ft["group"])) == {0, 1, 2}
    assert set(np.unique(ft["cycleskip_status"])) == {0, 1, 2, 3}
    assert (res.flags & S.FLAG_HPOL_RUN).any() and (res.flags & S.FLAG_COHORT_FP).any()
    assert ft["hmer_indel_length"].max() >= 15


def test_pileup_oracle_small_known_answer():
    # locus 0: 3 ref fwd (bq 10,20,30), 1 alt rev (bq 40); locus 1: empty; locus 2: 2 alt fwd, 1 other
    enc = lambda a, s, q: a | (s << 2) | (q << 3)
    obs = np.array([enc(0, 0, 10), enc(0, 0, 20), enc(0, 0, 30), enc(1, 1, 40), enc(1, 0, 7), enc(1, 0, 8), enc(2, 1, 9)],
                   np.uint16)
    off = np.array([0, 4, 4, 7], np.int64)
    r = O.pileup_tally(off, obs)
    assert r["ref_fwd"].tolist() == [3, 0, 0] and r["alt_rev"].tolist() == [1, 0, 0]
    assert r["alt_fwd"].tolist() == [0, 0, 2] and r["other"].tolist() == [0, 0, 1]
    assert r["dp"].tolist() == [4, 0, 3] and r["bq_ref"].tolist() == [60, 0, 0] and r["bq_alt"].tolist() == [40, 0, 15]
    assert r["vaf"].tolist() == [0.25, 0.0, np.float32(2) / np.float32(3)]
    # GATK StrandOddsRatio on the +1 table, locus 0: [[4,1],[1,2]]
    R = (4 * 2) / (1 * 1)
    assert np.isclose(r["sor64"][0], np.log(R + 1 / R) + np.log(1 / 4) - np.log(1 / 2))


def test_oracle_spec_matches_the_product_schema():
    """The oracle restates every encoding for itself (oracle/spec.py, with the reference lines it follows) instead of
    importing the product's constants - a slip in `variantcalling_amd/schema.py` can no longer cancel out between the two
    sides of a parity test.  This is the ONE place the two statements are compared, name by name."""
    from oracle import spec as P
    from variantcalling_amd import schema as S
    names = [n for n in dir(P) if n.isupper() and not n.startswith("_")]
    assert {"BASE_FEATURES", "FLAG_HPOL_RUN", "FLAG_COHORT_FP", "FLAG_SEC", "FLAG_TRACK0_SHIFT", "FILTER_LOW_SCORE", "GROUP_NAMES",
            "MODEL_GBT", "CSS_NA", "INDEL_DEL", "GC_WINDOW", "MOTIF_SIZE", "N_BASE_FEATURES", "MAX_TRACKS"} <= set(names)
    for n in names:
        assert getattr(P, n) == getattr(S, n), n
    assert P.N_BASE_FEATURES == len(P.BASE_FEATURES)
    seq = "ACGTNacgtnRYK-"
    assert np.array_equal(P.encode_bases(seq), S.encode_bases(seq))
    # and the oracle modules take their numbers from the spec, not from the product
    import ast
    import inspect
    from oracle import idiom, oracle
    for mod in (oracle, idiom):
        tree = ast.parse(inspect.getsource(mod))
        used = {node.attr for node in ast.walk(tree) if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id == "S"}
        assert used <= {"Reference", "IntervalTrack", "VariantTable", "FlatForest", "FilterResult"}, (mod.__name__, used)
