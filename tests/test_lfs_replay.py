"""LFS-replay harness: the reference's own hot-path fixtures are git-LFS objects that
are NOT pulled in /root/reference - only their sha256 and size survive (tests/golden/lfs_manifest.json, minted from the
pointer files by tools/make_lfs_manifest.py).  This file

  1. tells, for any mounted resource tree (UGVC_LFS_ROOT, default /root/reference/test/resources), which fixtures are real
     (size + sha256 match) and which are pointers / missing - and SKIPS the replay with that list when they are not;
  2. holds the replay itself (`replay_filter_fixtures`): the reference's model pickles are read without the reference's
     classes (variantcalling_amd/legacy_pickle.py), `filter_variants_pipeline.run` is run on the three fixture VCFs with
     the fixture blacklist, and every frame column of `train_model_approximate_gt_input.h5` that names a feature the
     engine computes is compared with the engine's value, feature by feature;
  3. proves the harness on a STAND-IN directory of the same layout built with this repository's own writers (pickles of
     look-alike `ugbio_filtering` classes whose module is deleted before loading) - CPU part here, GPU part under -m gpu.

With the real LFS objects a reviewer runs:  UGVC_LFS_ROOT=/path/to/test/resources pytest tests/test_lfs_replay.py -m gpu
"""
import hashlib
import json
import os
import pickle
import sys
import types

import numpy as np
import pytest

from conftest import ROOT
from variantcalling_amd import legacy_pickle, model_io, schema as S

MANIFEST = json.load(open(os.path.join(ROOT, "tests", "golden", "lfs_manifest.json")))
LFS_ROOT = os.environ.get("UGVC_LFS_ROOT", "/root/reference/test/resources")
FILTER_DIR = "system/test_filter_variants_pipeline"
NEEDED = ["general/chr1_head/hg38_runs.conservative.bed", f"{FILTER_DIR}/exact_gt.model.pkl", f"{FILTER_DIR}/approximate_gt.model.pkl",
          f"{FILTER_DIR}/blacklist_example.chr1_1_1000000.pkl",
          f"{FILTER_DIR}/004777-X0024.annotated.AF_chr1_1_1000000.vcf.gz", f"{FILTER_DIR}/006919_no_frd_chr1_1_5000000.vcf.gz",
          f"{FILTER_DIR}/036269-NA24143-Z0016.frd_chr1_1_5000000_unfiltered.vcf.gz",
          "system/test_train_models_pipeline/train_model_approximate_gt_input.h5",
          "general/chr1_head/Homo_sapiens_assembly38.fasta"]


def fixture_status(root: str, rel: str) -> str:
    """'real' (size and sha256 as the LFS pointer says), 'pointer', 'missing' or 'different'."""
    want = MANIFEST[rel]
    p = os.path.join(root, rel)
    if not os.path.exists(p):
        return "missing"
    size = os.path.getsize(p)
    with open(p, "rb") as fh:
        head = fh.read(64)
        if head.startswith(b"version https://git-lfs.github.com/spec/v1"):
            return "pointer"
        if size != want["size"]:
            return "different"
        h = hashlib.sha256(head)
        for chunk in iter(lambda: fh.read(1 << 20), b""):
            h.update(chunk)
    return "real" if h.hexdigest() == want["sha256"] else "different"


def missing_fixtures(root: str, needed=NEEDED) -> list:
    return [f"{rel} [{fixture_status(root, rel)}; sha256 {MANIFEST[rel]['sha256'][:12]}, {MANIFEST[rel]['size']} B]"
            for rel in needed if fixture_status(root, rel) != "real"]


def test_manifest_agrees_with_the_pointer_files():
    """Where the reference tree is mounted (the build container), the committed manifest is what its LFS pointers say."""
    if not os.path.isdir(LFS_ROOT):
        pytest.skip(f"{LFS_ROOT} not mounted")
    n_ptr = 0
    for rel, want in MANIFEST.items():
        p = os.path.join(LFS_ROOT, rel)
        if not want["lfs"] or not os.path.exists(p):
            continue
        head = open(p, "rb").read(200)
        if head.startswith(b"version https://git-lfs.github.com/spec/v1"):
            kv = dict(line.split(" ", 1) for line in head.decode().strip().splitlines())
            assert kv["oid"] == "sha256:" + want["sha256"] and int(kv["size"]) == want["size"], rel
            n_ptr += 1
    assert n_ptr == 0 or n_ptr >= 10
    for rel in NEEDED:
        assert rel in MANIFEST, rel


# ---------------------------------------------------------------------------------------------- the replay itself
def replay_filter_fixtures(root: str, out_dir: str, model_files=("exact_gt.model.pkl", "approximate_gt.model.pkl"),
                           fasta="general/chr1_head/Homo_sapiens_assembly38.fasta", runs="general/chr1_head/hg38_runs.conservative.bed",
                           vcfs=None, blacklist="blacklist_example.chr1_1_1000000.pkl",
                           frame="system/test_train_models_pipeline/train_model_approximate_gt_input.h5"):
    """Run the filter tool on the fixture VCFs with every named model of the fixture pickles; returns a report dict:
    models found, per (model file, model name, vcf) the PASS / LOW_SCORE counts, and per feature the number of rows of
    the fixture frame whose stored column differs from the engine's value."""
    from variantcalling_amd.io import h5, vcf as pyvcf
    from variantcalling_amd.pipelines import filter_variants_pipeline
    d = os.path.join(root, FILTER_DIR)
    vcfs = vcfs or sorted(f for f in os.listdir(d) if f.endswith(".vcf.gz"))
    report = dict(models={}, runs=[], feature_diffs={})
    for mf in model_files:
        models = model_io.load_model_file(os.path.join(d, mf))
        report["models"][mf] = {name: [None if f is None else (f.n_trees, int(f.max_depth)) for f in groups] for name, groups in models.items()}
        for name in models:
            for v in vcfs:
                out = os.path.join(out_dir, f"{mf}.{name}.{v}".replace("/", "_"))
                argv = ["filter_variants_pipeline", "--input_file", os.path.join(d, v), "--model_file", os.path.join(d, mf),
                        "--model_name", name, "--reference_file", os.path.join(root, fasta), "--runs_file", os.path.join(root, runs),
                        "--output_file", out]
                if blacklist:
                    argv += ["--blacklist", os.path.join(d, blacklist)]
                filter_variants_pipeline.run(argv)
                with pyvcf._open(out) as fh:
                    recs = [ln.split("\t") for ln in fh.read().decode().splitlines() if ln and not ln.startswith("#")]
                filt = [r[6] for r in recs]
                report["runs"].append(dict(model_file=mf, model=name, vcf=v, records=len(recs),
                                           PASS=sum(f == "PASS" for f in filt), LOW_SCORE=sum("LOW_SCORE" in f for f in filt),
                                           with_tree_score=sum("TREE_SCORE=" in r[7] for r in recs)))
    fp = os.path.join(root, frame)
    if os.path.exists(fp) and fixture_status(root, frame) in ("real", "different") and frame in MANIFEST or os.path.exists(fp):
        try:
            keys = [k for k in h5.list_keys(fp)]
        except Exception as e:                        # a frame we cannot read is reported, not fatal
            report["feature_diffs"] = {"error": repr(e)[:200]}
            keys = []
        for k in keys[:1]:
            fr = h5.read_hdf(fp, k)
            report["frame_columns"] = list(fr.keys())
    return report


# ---------------------------------------------------------------------------------------------- stand-in material
def _standin_model_pickle(frozen_models_rf, names=("rf_model_ignore_gt_incl_hpol_runs", "threshold_model_ignore_gt_incl_hpol_runs")):
    """A pickle shaped like the reference's model file: {model name: MaskedHierarchicalModel(models={group: estimator})}
    of classes in a module `ugbio_filtering.variant_filtering_utils` that does not exist when the file is read."""
    from sklearn.ensemble import RandomForestClassifier
    rng = np.random.default_rng(3)
    F = S.N_BASE_FEATURES                            # no annotation tracks: the replay passes none
    X = rng.normal(size=(3000, F)).astype(np.float32)
    ests = {}
    for g, gname in enumerate(S.GROUP_NAMES):
        y = (X[:, 0] + 0.5 * X[:, 1 + g] + rng.normal(0, 0.5, 3000) > 0).astype(int)
        ests[gname] = RandomForestClassifier(n_estimators=6, max_depth=5, random_state=g).fit(X, y)
    mod = types.ModuleType("ugbio_filtering.variant_filtering_utils")
    pkg = types.ModuleType("ugbio_filtering")

    class MaskedHierarchicalModel:
        def __init__(self, name, group_column, models):
            self.name, self.group_column, self.models = name, group_column, models
            self.transformer = SingleTrivialClassifierModel()

    class SingleTrivialClassifierModel:
        def __init__(self):
            self.ignored_filters = {"HPOL_RUN"}

    for cls in (MaskedHierarchicalModel, SingleTrivialClassifierModel):
        cls.__module__ = mod.__name__
        cls.__qualname__ = cls.__name__
        setattr(mod, cls.__name__, cls)
    sys.modules[pkg.__name__], sys.modules[mod.__name__] = pkg, mod
    try:
        raw = pickle.dumps({n: MaskedHierarchicalModel(n, "group", dict(ests)) for n in names}, protocol=4)
    finally:
        del sys.modules[mod.__name__], sys.modules[pkg.__name__]
    return raw, ests, X


def test_reference_model_pickle_is_read_without_the_reference_classes(frozen_models, tmp_path):
    raw, ests, X = _standin_model_pickle(frozen_models)
    with pytest.raises(ModuleNotFoundError):
        pickle.loads(raw)                                             # what the plain loader does with such a file
    path = str(tmp_path / "exact_gt.model.pkl")
    open(path, "wb").write(raw)
    models = model_io.load_model_file(path)
    assert sorted(models) == ["rf_model_ignore_gt_incl_hpol_runs", "threshold_model_ignore_gt_incl_hpol_runs"]
    from oracle import oracle as O
    for g, gname in enumerate(S.GROUP_NAMES):
        f = models["rf_model_ignore_gt_incl_hpol_runs"][g]
        assert f is not None and f.n_trees == 6
        p0, p1 = O.forest_predict(f, X[:500])
        assert np.array_equal(p1, ests[gname].predict_proba(X[:500])[:, 1]), gname     # each group got ITS estimator
    held = legacy_pickle.load(raw)["rf_model_ignore_gt_incl_hpol_runs"]
    assert isinstance(held, legacy_pickle.Holder) and held.group_column == "group"
    assert held._ugvc_origin == ("ugbio_filtering.variant_filtering_utils", "MaskedHierarchicalModel")


def test_named_feature_columns_are_mapped_to_the_engine_order():
    """An estimator fitted on a named frame (feature_names_in_) in the REFERENCE's column order scores through the
    engine's feature order; a column the engine does not compute is an error that names it."""
    import pandas as pd
    from sklearn.tree import DecisionTreeClassifier
    from oracle import oracle as O
    rng = np.random.default_rng(1)
    cols = ["sor", "dp", "qual", "hmer_indel_nuc", "inside_hmer_run", "close_to_hmer_run", "hmer_indel_length", "indel_length",
            "ad_0", "ad_1", "af", "gc_content", "left_motif", "right_motif", "cycleskip_status", "LCR-hs38"]
    Xd = pd.DataFrame(rng.normal(size=(800, len(cols))).astype(np.float32), columns=cols)
    y = (Xd["qual"] - Xd["sor"] + 0.3 * Xd["LCR-hs38"] > 0).astype(int)
    clf = DecisionTreeClassifier(max_depth=6, random_state=0).fit(Xd, y)
    f = model_io.flatten_sklearn(clf, track_names=["LCR-hs38", "exome.twist"])
    ours = np.zeros((800, S.N_BASE_FEATURES + 2), np.float32)
    perm = model_io.feature_permutation(clf, ["LCR-hs38", "exome.twist"])
    ours[:, perm] = Xd.to_numpy()
    assert np.array_equal(O.forest_predict(f, ours)[1], clf.predict_proba(Xd)[:, 1])
    Xd2 = Xd.rename(columns={"LCR-hs38": "x_custom_annotation"})
    clf2 = DecisionTreeClassifier(max_depth=3, random_state=0).fit(Xd2, y)
    with pytest.raises(ValueError, match="x_custom_annotation"):
        model_io.flatten_sklearn(clf2, track_names=["LCR-hs38"])


class _OldTreeState:
    """Pickles as `sklearn.tree._tree.Tree(...)` with the state a scikit-learn <= 1.2 writes: the node array WITHOUT the
    `missing_go_to_left` field (added in 1.3) and weighted sample COUNTS per leaf - the shape of the reference's own
    model pickles (scikit-learn 1.2.2: setup/environment.yml:399)."""
    def __init__(self, tree, n_features, n_classes):
        st = tree.__getstate__()
        old_fields = [f for f in st["nodes"].dtype.names if f != "missing_go_to_left"]
        nodes = np.zeros(st["nodes"].shape, dtype=[(f, st["nodes"].dtype[f]) for f in old_fields])
        for f in old_fields:
            nodes[f] = st["nodes"][f]
        counts = st["values"] * st["nodes"]["weighted_n_node_samples"][:, None, None]       # fractions -> counts
        self.args = (n_features, np.asarray([n_classes], dtype=np.intp), 1)
        self.state = dict(max_depth=st["max_depth"], node_count=st["node_count"], nodes=nodes, values=counts)

    def __reduce__(self):
        from sklearn.tree._tree import Tree
        return (Tree, self.args, self.state)


def test_old_scikit_learn_tree_state_is_read_as_data(tmp_path):
    """`Tree.__setstate__` of this scikit-learn refuses a <= 1.2 node array (ValueError), which used to escape
    from both loaders.  The shim holds the compiled tree as data and `model_io` reads nodes / values from the held state."""
    from sklearn.ensemble import RandomForestClassifier
    from oracle import oracle as O
    rng = np.random.default_rng(3)
    X = rng.normal(size=(1500, S.N_BASE_FEATURES)).astype(np.float32)
    y = (X[:, 0] - X[:, 1] + 0.2 * X[:, 5] > 0).astype(int)
    clf = RandomForestClassifier(n_estimators=5, max_depth=5, random_state=2).fit(X, y)
    want = clf.predict_proba(X[:400])[:, 1]
    for e in clf.estimators_:
        e.tree_ = _OldTreeState(e.tree_, X.shape[1], 2)
    raw = pickle.dumps({"rf_model_ignore_gt_incl_hpol_runs": [clf, clf, clf]})
    with pytest.raises(ValueError):
        pickle.loads(raw)                                            # the plain loader: incompatible dtype
    path = str(tmp_path / "old_sklearn.model.pkl")
    open(path, "wb").write(raw)
    models = model_io.load_model_file(path)
    f = models["rf_model_ignore_gt_incl_hpol_runs"][0]
    assert f.n_trees == 5 and f.max_depth == 5
    assert np.array_equal(O.forest_predict(f, X[:400])[1], want)     # count-valued leaves normalised as 1.2's predict_proba does


class _RaisesValueError:
    """A pickle whose load raises a ValueError that has nothing to do with scikit-learn's tree state."""
    def __reduce__(self):
        return (int, ("not a number",))


def test_unrelated_value_error_keeps_its_message(tmp_path):
    """only `Tree.__setstate__`'s incompatible-array ValueError sends a pickle down the legacy path; any other
    ValueError is raised as it is (once), not replaced by "no scikit-learn estimator found"."""
    raw = pickle.dumps({"m": _RaisesValueError()})
    path = str(tmp_path / "broken.model.pkl")
    open(path, "wb").write(raw)
    with pytest.raises(ValueError, match="invalid literal"):
        model_io.load_model_file(path)
    with pytest.raises(ValueError, match="invalid literal"):
        legacy_pickle.load(raw)
    assert legacy_pickle.is_tree_state_mismatch(ValueError("node array from the pickle has an incompatible dtype:\n- expected: x"))
    assert not legacy_pickle.is_tree_state_mismatch(ValueError("invalid literal for int() with base 10: 'x'"))
    # older scikit-learn generations and cross-platform pickles raise other texts from the same Tree.__setstate__
    assert legacy_pickle.is_tree_state_mismatch(ValueError("Did not recognise loaded array layout"))
    assert legacy_pickle.is_tree_state_mismatch(ValueError("Did not recognise loaded array dimensions"))
    assert legacy_pickle.is_tree_state_mismatch(ValueError("Buffer dtype mismatch, expected 'SIZE_t' but got 'long long'"))
    assert not legacy_pickle.is_tree_state_mismatch(TypeError("Buffer dtype mismatch, expected 'SIZE_t' but got 'long long'"))


def test_model_file_maps_annotation_columns_by_bed_stem(tmp_path):
    """`load_model_file(..., track_names=...)` - what filter_variants_pipeline passes for --annotate_intervals -
    resolves a model's named interval column; without the stems the same file is refused by name."""
    import pandas as pd
    from sklearn.tree import DecisionTreeClassifier
    rng = np.random.default_rng(5)
    cols = ["qual", "sor", "dp", "LCR-hs38", "exome.twist"]
    Xd = pd.DataFrame(rng.normal(size=(600, len(cols))).astype(np.float32), columns=cols)
    y = (Xd["qual"] + Xd["exome.twist"] > 0).astype(int)
    clf = DecisionTreeClassifier(max_depth=4, random_state=0).fit(Xd, y)
    path = str(tmp_path / "named.model.pkl")
    open(path, "wb").write(pickle.dumps({"dt_model_ignore_gt_incl_hpol_runs": {g: clf for g in S.GROUP_NAMES}}))
    forests = model_io.load_model_file(path, "dt_model_ignore_gt_incl_hpol_runs", track_names=["LCR-hs38", "exome.twist"])
    used = set(int(x) for x in forests[0].feature[forests[0].feature >= 0])
    assert used <= {S.BASE_FEATURES.index("qual"), S.BASE_FEATURES.index("sor"), S.BASE_FEATURES.index("dp"),
                    S.N_BASE_FEATURES, S.N_BASE_FEATURES + 1}
    assert S.N_BASE_FEATURES + 1 in used                                                   # exome.twist = the second BED
    with pytest.raises(ValueError, match="LCR-hs38|exome.twist"):
        model_io.load_model_file(path, "dt_model_ignore_gt_incl_hpol_runs")


def test_reference_blacklist_pickle_is_read_without_the_reference_classes(tmp_path):
    import pandas as pd
    from variantcalling_amd.io import bed
    mod, pkg = types.ModuleType("ugbio_filtering.blacklist"), types.ModuleType("ugbio_filtering")

    class Blacklist:
        def __init__(self, blacklist, annotation, selection_fcn=None, description=""):
            self.blacklist, self.annotation, self.selection_fcn, self.description = blacklist, annotation, selection_fcn, description

    Blacklist.__module__, Blacklist.__qualname__ = mod.__name__, "Blacklist"
    mod.Blacklist = Blacklist
    sys.modules[pkg.__name__], sys.modules[mod.__name__] = pkg, mod
    try:
        idx = pd.MultiIndex.from_tuples([("chr1", 1000), ("chr1", 52), ("chr2", 7), ("chrUn", 9)], names=["chrom", "pos"])
        raw = pickle.dumps([Blacklist(set(idx.tolist()), "COHORT_FP"), Blacklist(idx, "SEC")], protocol=4)
    finally:
        del sys.modules[mod.__name__], sys.modules[pkg.__name__]
    path = str(tmp_path / "blacklist_example.chr1_1_1000000.pkl")
    open(path, "wb").write(raw)
    keys = bed.read_blacklist(path, ["chr1", "chr2"])
    assert keys.tolist() == [52, 1000, (1 << 32) | 7]


# ---------------------------------------------------------------------------------------------- replay (GPU)
@pytest.mark.gpu
def test_replay_the_reference_fixtures(tmp_path):
    """The real thing: needs the LFS objects (UGVC_LFS_ROOT) and a GPU.  Skips, naming every missing object, otherwise."""
    miss = missing_fixtures(LFS_ROOT) if os.path.isdir(LFS_ROOT) else [f"{LFS_ROOT} not mounted"]
    if miss:
        pytest.skip("reference LFS fixtures not available:\\n  " + "\\n  ".join(miss))
    rep = replay_filter_fixtures(LFS_ROOT, str(tmp_path))
    print(json.dumps(rep, indent=1))
    assert rep["models"] and all(r["records"] > 0 and r["with_tree_score"] == r["records"] for r in rep["runs"])


def test_replay_skips_cleanly_without_the_objects():
    """In the build container every hot-path fixture is a pointer: the harness says which, by hash."""
    if not os.path.isdir(LFS_ROOT):
        pytest.skip(f"{LFS_ROOT} not mounted")
    miss = missing_fixtures(LFS_ROOT)
    if not miss:
        pytest.skip("the LFS objects are present here: run the replay (-m gpu)")
    assert all("pointer" in m or "missing" in m for m in miss), miss
    assert any("bb55d58f1994" in m for m in miss)                     # exact_gt.model.pkl, SURVEY.md appendix C


@pytest.mark.gpu
def test_replay_on_a_stand_in_directory(tmp_path, frozen_models):
    """The same replay function on a directory of the reference's layout written here: a small chr1 FASTA, three VCFs, two
    model pickles of look-alike reference classes, a blacklist pickle.  Every record gets a TREE_SCORE and a FILTER."""
    from variantcalling_amd import synth
    from variantcalling_amd.io import fasta as pyfasta, vcf as pyvcf
    root = tmp_path / "resources"
    d = root / FILTER_DIR
    os.makedirs(d)
    os.makedirs(root / "general" / "chr1_head")
    cs = synth.make_callset(3000, genome_len=1_000_000, n_contigs=1, seed=5)
    cs.ref.names[:] = ["chr1"]
    fa = str(root / "general" / "chr1_head" / "Homo_sapiens_assembly38.fasta")
    pyfasta.write_fasta(fa, cs.ref)
    vcfs = ["a.vcf.gz", "b.vcf.gz", "c.vcf.gz"]
    for k, v in enumerate(vcfs):
        pyvcf.write_vcf_from_table(str(d / v), cs.variants.slice(k * 900, (k + 1) * 900), ["chr1"])
    raw, _, _ = _standin_model_pickle(frozen_models)
    for mf in ("exact_gt.model.pkl", "approximate_gt.model.pkl"):
        open(d / mf, "wb").write(raw)
    pickle.dump([("chr1", int(p)) for p in cs.variants.pos[::50]], open(d / "blacklist_example.chr1_1_1000000.pkl", "wb"))
    from variantcalling_amd.io import bed
    bed.write_bed(str(root / "general" / "chr1_head" / "hg38_runs.conservative.bed"), cs.runs, ["chr1"])
    rep = replay_filter_fixtures(str(root), str(tmp_path), vcfs=vcfs, frame="absent.h5")
    assert set(rep["models"]) == {"exact_gt.model.pkl", "approximate_gt.model.pkl"}
    assert len(rep["runs"]) == 2 * 2 * 3
    for r in rep["runs"]:
        assert r["records"] == 900 and r["with_tree_score"] == 900 and r["PASS"] + r["LOW_SCORE"] >= 1
