"""GPU end-to-end tests of the drop-in tools (config C1: a chr20-shaped 50 k-variant VCF through
filter_variants_pipeline with a pre-trained model; and train_models_pipeline -> filter_variants_pipeline)."""
import gzip
import os

import numpy as np
import pytest

from conftest import GOLDEN
from variantcalling_amd import model_io, schema as S, synth
from variantcalling_amd.io import bed, fasta, vcf as vcfio

pytestmark = pytest.mark.gpu
RF = "rf_model_ignore_gt_incl_hpol_runs"


def _write_inputs(tmp_path, cs, ids=None, gz=True):
    d = {}
    d["fa"] = str(tmp_path / "ref.fa"); fasta.write_fasta(d["fa"], cs.ref)
    d["vcf"] = str(tmp_path / ("calls.vcf.gz" if gz else "calls.vcf"))
    vcfio.write_vcf_from_table(d["vcf"], cs.variants, cs.ref.names, ids=ids)
    d["runs"] = str(tmp_path / "runs.bed"); bed.write_bed(d["runs"], cs.runs, cs.ref.names)
    d["ann"] = []
    for t, tr in zip(("LCR-hs38", "exome.twist", "mappability.0"), cs.tracks):
        p = str(tmp_path / f"{t}.bed"); bed.write_bed(p, tr, cs.ref.names); d["ann"] += ["--annotate_intervals", p]
    d["bl"] = str(tmp_path / "blacklist.npy"); np.save(d["bl"], cs.blacklist)
    return d


def _parse_out(path, n):
    op = gzip.open if path.endswith(".gz") else open
    recs = [x.split("\t") for x in op(path, "rt").read().splitlines() if not x.startswith("#")]
    assert len(recs) == n
    score = np.array([np.float32(dict(kv.split("=") for kv in r[7].split(";") if "=" in kv)["TREE_SCORE"]) for r in recs])
    tags = [set(r[6].split(";")) for r in recs]
    return score, tags


def test_c1_filter_variants_pipeline_chr20_50k(tmp_path, frozen_models):
    from oracle import oracle as O
    from variantcalling_amd.pipelines import filter_variants_pipeline
    cs = synth.make_callset(50_000, genome_len=64_444_167, n_contigs=1, seed=20)      # one chr20-sized contig
    cs.genome.ref.names[:] = ["chr20"]
    d = _write_inputs(tmp_path, cs)
    out = str(tmp_path / "filtered.vcf.gz")
    rc = filter_variants_pipeline.run(["filter_variants_pipeline", "--input_file", d["vcf"], "--model_file",
                                       os.path.join(GOLDEN, "synth_rf_v1.npz"), "--model_name", RF, "--runs_file", d["runs"],
                                       "--hpol_filter_length_dist", "10", "10", "--blacklist", d["bl"], "--reference_file",
                                       d["fa"], "--flow_order", "TGCA", "--output_file", out] + d["ann"])
    assert rc == 0
    exp = O.filter_variants(cs.variants, cs.ref, cs.runs, cs.tracks, cs.blacklist, frozen_models[RF])
    score, tags = _parse_out(out, cs.variants.n)
    assert np.array_equal(score, exp.tree_score)
    low = np.array(["LOW_SCORE" in t for t in tags]); hp = np.array(["HPOL_RUN" in t for t in tags])
    fp = np.array(["COHORT_FP" in t for t in tags]); ps = np.array([t == {"PASS"} for t in tags])
    assert np.array_equal(low, exp.filter == S.FILTER_LOW_SCORE)
    assert np.array_equal(hp, (exp.flags & S.FLAG_HPOL_RUN) > 0) and np.array_equal(fp, (exp.flags & S.FLAG_COHORT_FP) > 0)
    assert np.array_equal(ps, (exp.filter == 0) & (exp.flags & 3 == 0))
    assert 0.2 < ps.mean() < 0.8 and fp.sum() > 0 and hp.sum() > 0


def test_train_then_filter(tmp_path):
    """Approximate-ground-truth training (dbSNP id => TP, blacklist => FP) on the GPU feature matrix, then the
    trained pickle drives filter_variants_pipeline; the result equals the oracle evaluated with the same
    scikit-learn estimators."""
    import pickle

    from oracle import oracle as O
    from variantcalling_amd.pipelines import filter_variants_pipeline, train_models_pipeline
    cs = synth.make_callset(30_000, genome_len=20_000_000, n_contigs=2, seed=33)
    rng = np.random.default_rng(3)
    # labels correlated with qual so the forests have something to learn
    is_tp = (cs.variants.qual + rng.normal(0, 25, cs.variants.n)) > 45
    bl = np.unique(cs.variants.keys()[~is_tp & (rng.random(cs.variants.n) < 0.8)])
    cs.blacklist = bl
    d = _write_inputs(tmp_path, cs, ids=is_tp & (rng.random(cs.variants.n) < 0.8), gz=False)
    prefix = str(tmp_path / "test.model")
    rc = train_models_pipeline.run(["train_models_pipeline", "--input_file", d["vcf"], "--reference", d["fa"],
                                    "--runs_intervals", d["runs"], "--blacklist", d["bl"], "--flow_order", "TGCA",
                                    "--exome_weight", "100", "--exome_weight_annotation", "exome.twist",
                                    "--output_file_prefix", prefix, "--evaluate_concordance"] + d["ann"])
    assert rc == 0
    models = pickle.load(open(prefix + ".pkl", "rb"))
    assert set(models) == {"rf_model_ignore_gt_incl_hpol_runs", "dt_model_ignore_gt_incl_hpol_runs",
                           "rf_model_ignore_gt_excl_hpol_runs", "dt_model_ignore_gt_excl_hpol_runs",
                           "threshold_model_ignore_gt_incl_hpol_runs", "threshold_model_ignore_gt_excl_hpol_runs",
                           "xgb_model_ignore_gt_incl_hpol_runs", "xgb_model_ignore_gt_excl_hpol_runs"}
    rows = [x.split(";") for x in open(prefix + ".stats.csv").read().splitlines()]
    assert rows[0][0] == "group" and rows[1][0] == "SNP" and float(rows[1][6]) > 0.7      # SNP f1 after filtering
    from variantcalling_amd.io import h5
    res = h5.read_hdf(prefix + ".h5", "training_set")
    ft = O.featurize(cs.variants, cs.ref, cs.runs, cs.tracks)
    feat = S.feature_names(len(cs.tracks))
    assert list(res.keys()) == ["chrom", "pos", "label", "group"] + list(feat)
    assert np.array_equal(np.stack([res[f] for f in feat], axis=1), ft["X"]) and np.array_equal(res["group"], ft["group"])
    assert np.array_equal(res["pos"], cs.variants.pos)
    scored = h5.read_hdf(prefix + ".h5", "scored_concordance")
    acc = h5.read_hdf(prefix + ".h5", "optimal_recall_precision")
    assert [str(x) for x in acc["tp"]] == [r[1] for r in rows[1:]] and list(acc["group"]) == [r[0] for r in rows[1:]]
    assert scored.n_rows == cs.variants.n and set(scored["filter"]) <= {"PASS", "LOW_SCORE", "HPOL_RUN", "COHORT_FP", "HPOL_RUN;LOW_SCORE",
                                                                     "COHORT_FP;LOW_SCORE", "HPOL_RUN;COHORT_FP", "HPOL_RUN;COHORT_FP;LOW_SCORE"}
    # exact-label mode: the scored frame (classify = tp / fp) is itself a comparison HDF5; training on it sees the same
    # labels and builds the same feature matrix
    h5.write_hdf(str(tmp_path / "labelled.h5"), {"chr_a": scored})
    prefix2 = str(tmp_path / "test.model2")
    rc = train_models_pipeline.run(["train_models_pipeline", "--input_file", str(tmp_path / "labelled.h5"), "--reference", d["fa"],
                                    "--runs_intervals", d["runs"], "--flow_order", "TGCA", "--output_file_prefix", prefix2,
                                    "--ignore_filter_status",                     # (the frame carries round one's FILTER tags)
                                    # config C5 inside the tool: the gradient-boosted ensemble applied by the scoring pass AND as a
                                    # leaf-matrix GEMM on the resident feature matrix - the tool raises if the two disagree
                                    "--evaluate_concordance", "--apply_model", "xgb_model_ignore_gt_incl_hpol_runs"] + d["ann"])
    assert rc == 0
    assert h5.read_hdf(prefix2 + ".h5", "scored_concordance").n_rows == cs.variants.n
    res2 = h5.read_hdf(prefix2 + ".h5", "training_set")
    assert np.array_equal(res2["label"], res["label"]) and np.array_equal(res2["pos"], res["pos"])
    for f in feat:
        assert np.array_equal(res2[f], res[f]), f
    for name in ("dt_model_ignore_gt_excl_hpol_runs", "rf_model_ignore_gt_incl_hpol_runs", "threshold_model_ignore_gt_incl_hpol_runs",
                 "xgb_model_ignore_gt_incl_hpol_runs"):
        out = str(tmp_path / f"{name}.vcf")
        filter_variants_pipeline.run(["filter_variants_pipeline", "--input_file", d["vcf"], "--model_file", prefix + ".pkl",
                                      "--model_name", name, "--runs_file", d["runs"], "--blacklist", d["bl"],
                                      "--reference_file", d["fa"], "--output_file", out] + d["ann"])
        forests = [(models[name][g] if isinstance(models[name][g], S.FlatForest) else model_io.flatten_sklearn(models[name][g]))
                   if g in models[name] else None for g in S.GROUP_NAMES]
        exp = O.filter_variants(cs.variants, cs.ref, cs.runs, cs.tracks, bl, forests)
        score, tags = _parse_out(out, cs.variants.n)
        g0 = ft["group"] == 0
        if name.startswith("xgb"):
            # the gradient-boosted ensemble (round 6: fitted by scikit-learn's histogram gradient boosting, stored and scored in
            # XGBoost's format): FILTER decided on the bit-exact f32 margin; TREE_SCORE = sigmoid through device libm (1e-6 abs,
            # the stated tolerance of the GBT path); against scikit-learn's own f64 probabilities to 1e-5
            assert all(f.kind == S.MODEL_GBT and f.n_trees == 100 and f.max_depth <= 6 for f in forests)
            assert np.array_equal(np.array(["LOW_SCORE" in t for t in tags]), exp.filter == 1)
            assert np.abs(score - exp.tree_score).max() <= 1e-6
            sk = models[name]["snp"].predict_proba(ft["X"][g0])[:, 1]
            assert np.abs(exp.tree_score[g0] - sk).max() < 1e-5 and ((exp.tree_score[g0] > 0.5) == is_tp[g0]).mean() > 0.75
            continue
        assert np.array_equal(score, exp.tree_score)
        assert np.array_equal(np.array(["LOW_SCORE" in t for t in tags]), exp.filter == 1)
        if name.startswith("threshold"):                       # the two-feature model: labels were drawn from QUAL, so it must separate
            assert ((exp.tree_score[g0] > 0.5) == is_tp[g0]).mean() > 0.75
            continue
        # and against scikit-learn itself
        assert np.array_equal(exp.tree_score[g0], models[name]["snp"].predict_proba(ft["X"][g0])[:, 1].astype(np.float32))


def test_wgs_sized_contig_list_and_multiallelic_records_through_the_cli(tmp_path, frozen_models):
    """A reference with the 3 366 contigs of the reference's production header (tests/golden/hg38_contigs.tsv.gz; short
    synthetic sequences under the real names): calls on contigs beyond index 255 (alt / decoy / HLA) go through the tool,
    multi-allelic records are scored per ALT allele and folded into one FILTER / TREE_SCORE per record (io/multiallelic.py),
    a spanning-deletion allele gets no row.  Expected values: the oracle on the expanded table, folded by the same rule."""
    from oracle import oracle as O
    from variantcalling_amd.io import multiallelic, vcf_native
    from variantcalling_amd.pipelines import filter_variants_pipeline
    names = [ln.split("\t")[0] for ln in gzip.open(os.path.join(GOLDEN, "hg38_contigs.tsv.gz"), "rt") if not ln.startswith("#")]
    rng = np.random.default_rng(12)
    L = 1500
    codes = rng.integers(1, 5, size=len(names) * L).astype(np.uint8)
    ref = S.Reference(codes, (np.arange(len(names) + 1) * L).astype(np.int64), list(names))
    fa = str(tmp_path / "wide.fa"); fasta.write_fasta(fa, ref)
    lines = ["##fileformat=VCFv4.2"] + [f"##contig=<ID={n},length={L}>" for n in names] + \
            ["#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ts1"]
    letters = "NACGT"
    n_rec = 0
    for c in sorted(set(rng.integers(0, len(names), 400).tolist()) | {0, 255, 256, 3365}):
        for pos in sorted(set(rng.integers(20, L - 60, 3).tolist())):
            r = letters[codes[c * L + pos - 1]]
            others = [b for b in "ACGT" if b != r]
            kind = int(rng.integers(0, 5))
            if kind == 0:
                alt, ad = f"{others[0]},{others[1]}", "7,9,11"
            elif kind == 1:
                alt, ad = f"*,{r}{others[2]}", "6,5,12"
            elif kind == 2:
                alt, ad = f"{others[0]},*,{r}AA", "4,8,3,10"
            else:
                alt, ad = others[int(rng.integers(0, 3))], "10,12"
            dp = sum(int(x) for x in ad.split(","))
            lines.append(f"{names[c]}\t{pos}\t.\t{r}\t{alt}\t{rng.exponential(60):.2f}\t.\tSOR={rng.lognormal(0, .7):.3f}\t"
                         f"GT:AD:DP:GQ\t0/1:{ad}:{dp}:{int(rng.integers(0, 99))}")
            n_rec += 1
    vcf_path = str(tmp_path / "wide.vcf")
    open(vcf_path, "w").write("\n".join(lines) + "\n")
    out = str(tmp_path / "wide.filtered.vcf.gz")
    runs = str(tmp_path / "runs.bed"); open(runs, "w").write("")
    rc = filter_variants_pipeline.run(["filter_variants_pipeline", "--input_file", vcf_path, "--model_file", os.path.join(GOLDEN, "synth_rf_v1.npz"),
                                       "--model_name", RF, "--runs_file", runs, "--reference_file", fa, "--output_file", out] +
                                      sum((["--annotate_intervals", runs] for _ in range(3)), []))
    assert rc == 0
    v = vcf_native.read_vcf(vcf_path, names)
    assert int(v.table.contig.max()) == 3365 and (v.n_alt > 1).sum() > 50
    table, base = multiallelic.expand(v)
    assert table.n > v.table.n
    empty = [S.IntervalTrack(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(len(names) + 1, np.int32), f"t{j}") for j in range(3)]
    runs_t = S.IntervalTrack(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(len(names) + 1, np.int32), "runs")
    exp = multiallelic.collapse(O.filter_variants(table, ref, runs_t, empty, None, frozen_models[RF]), base, v.table.n)
    score, tags = _parse_out(out, n_rec)
    inv = np.argsort(v.order)                                   # file order -> table row (the input is sorted: identity)
    assert np.array_equal(score[v.order], exp.tree_score) or np.array_equal(score, exp.tree_score[inv])
    low = np.array(["LOW_SCORE" in t for t in tags])
    assert np.array_equal(low[v.order], exp.filter == S.FILTER_LOW_SCORE)


def test_evaluate_concordance_device_flag_gives_identical_files(tmp_path):
    """`evaluate_concordance --device N`: the cumulative performance curves come from ugvc_pr_curve (hand-written radix
    sort + scan + f64 finish) - the files must equal the host run's byte for byte (ties, NaN scores, missing candidates)."""
    import filecmp
    from test_evaluate_concordance import _frame_with_expected_counts, _split_by_chrom
    from variantcalling_amd.io import h5
    from variantcalling_amd.pipelines import evaluate_concordance
    rng = np.random.default_rng(8)
    fr = _frame_with_expected_counts(rng)
    sc = np.asarray(fr["tree_score"], np.float64)
    sc[::7] = np.round(sc[::7], 1)                     # runs of equal scores: the tie rule (input order) matters
    fr["tree_score"] = sc
    src = str(tmp_path / "in.h5")
    h5.write_hdf(src, _split_by_chrom(fr))
    evaluate_concordance.run(["--input_file", src, "--output_prefix", str(tmp_path / "host")])
    evaluate_concordance.run(["--input_file", src, "--output_prefix", str(tmp_path / "gpu"), "--device", "0"])
    for ext in (".h5", ".stats.csv", ".thresholds.csv"):
        assert filecmp.cmp(str(tmp_path / "host") + ext, str(tmp_path / "gpu") + ext, shallow=False), ext
    perf = h5.read_hdf(str(tmp_path / "gpu") + ".h5", "performance_curve")
    assert perf["score"][0].size > 1000 and np.isfinite(perf["f1"][0]).any()


def test_sec_training_then_correct_systematic_errors(tmp_path):
    """The two SEC tools end to end (/root/reference/ugvc/__main__.py:19,56; flags BUILDER-DEFINED): a database from four
    cohort VCFs, applied to a fifth callset; database and verdicts equal oracle/stats.py on the same inputs."""
    import gzip
    from oracle import stats as st
    from variantcalling_amd.pipelines import correct_systematic_errors, sec_training
    cs = synth.make_callset(30_000, genome_len=20_000_000, n_contigs=3, seed=77)
    fa = str(tmp_path / "ref.fa"); fasta.write_fasta(fa, cs.ref)
    rng = np.random.default_rng(5)
    cohort, obs_k, obs_c = [], [], []
    for s in range(4):
        keep = np.flatnonzero(rng.random(cs.variants.n) < 0.6)
        parts = [cs.variants.slice(int(a), int(a) + 1) for a in keep[:0]]          # (slices are contiguous: build by mask below)
        vt = cs.variants
        sub = S.VariantTable(**{c: np.ascontiguousarray(getattr(vt, c)[keep]) for c in S.VariantTable.COLS if c not in ("ref_off", "alt_off")},
                             ref_off=vt.ref_off[keep], alt_off=vt.alt_off[keep], alleles=vt.alleles)
        sub.dp = rng.integers(10, 60, keep.size).astype(np.int32)
        sub.ad_alt = (sub.dp * rng.random(keep.size) * 0.6).astype(np.int32)
        sub.ad_ref = (sub.dp - sub.ad_alt - rng.integers(0, 3, keep.size)).clip(0).astype(np.int32)
        p = str(tmp_path / f"sample{s}.vcf.gz")
        vcfio.write_vcf_from_table(p, sub, cs.ref.names)
        cohort += ["--inputs", p]
        k, c = sec_training.observations(sub)
        obs_k.append(k); obs_c.append(c)
    db = str(tmp_path / "cohort.sec.npz")
    assert sec_training.run(["sec_training", "--reference_file", fa, "--output_file", db] + cohort) == 0
    z = np.load(db)
    ok, oe = st.sec_db_build(np.concatenate(obs_k), np.concatenate(obs_c))
    assert np.array_equal(z["keys"], ok) and np.array_equal(z["expected"], oe) and int(z["n_samples"]) == 4
    # apply to the full callset
    calls = str(tmp_path / "calls.vcf.gz")
    vcfio.write_vcf_from_table(calls, cs.variants, cs.ref.names)
    out = str(tmp_path / "calls.sec.vcf.gz")
    assert correct_systematic_errors.run(["correct_systematic_errors", "--input_file", calls, "--sec_db", db, "--reference_file", fa,
                                          "--output_file", out, "--min_ratio", "0.05"]) == 0
    vt = cs.variants
    keys = (vt.contig.astype(np.uint64) << np.uint64(32)) | vt.pos.astype(np.uint64)
    ratio, hit = st.sec_apply(keys, vt.dp, vt.ad_ref, vt.ad_alt, ok, oe, 0.05, True)
    lines = [x for x in gzip.open(out, "rt").read().splitlines()]
    assert any(x.startswith("##FILTER=<ID=SEC") for x in lines) and any(x.startswith("##INFO=<ID=SEC_LR") for x in lines)
    recs = [x.split("\t") for x in lines if not x.startswith("#")]
    assert len(recs) == vt.n
    got_hit = np.array(["SEC" in r[6].split(";") for r in recs])
    got_lr = np.array([float(dict(kv.split("=") for kv in r[7].split(";") if "=" in kv).get("SEC_LR", "nan")) for r in recs])
    on_db = ~np.isnan(ratio)
    assert np.array_equal(~np.isnan(got_lr), on_db) and on_db.sum() > 10_000
    sliver = np.abs(ratio - 0.05) < 1e-9                  # device lgamma / exp vs scipy: equal outside this sliver
    assert np.array_equal(got_hit[~sliver], hit[~sliver]) and 0 < hit.sum() < on_db.sum()
    assert np.allclose(got_lr[on_db], ratio[on_db].astype(np.float32), rtol=1e-6, atol=1e-30)
    assert os.path.exists(out + ".tbi")
    # ---- sec_validation (round 4): the database against the callset as a held-out sample - the consumer of the ratios
    import csv
    from variantcalling_amd.pipelines import sec_validation
    prefix = str(tmp_path / "val")
    assert sec_validation.run(["sec_validation", "--inputs", calls, "--inputs", str(tmp_path / "sample0.vcf.gz"), "--sec_db", db,
                               "--reference_file", fa, "--output_prefix", prefix, "--min_ratio", "0.05"]) == 0
    rows = list(csv.DictReader(open(prefix + ".sec_validation.csv")))
    assert [r["sample"] for r in rows] == [calls, str(tmp_path / "sample0.vcf.gz"), "ALL"]
    r0 = rows[0]
    assert int(r0["n_calls"]) == vt.n and int(r0["n_on_database"]) == int(on_db.sum())
    assert abs(int(r0["n_sec"]) - int(hit.sum())) <= int(sliver.sum())
    assert abs(float(r0["ratio_q50"]) - float(np.quantile(ratio[on_db], 0.5))) <= 1e-9 * max(1.0, float(np.quantile(ratio[on_db], 0.5)))
    assert int(rows[2]["n_calls"]) == int(rows[0]["n_calls"]) + int(rows[1]["n_calls"])
    thr = list(csv.DictReader(open(prefix + ".sec_validation.thresholds.csv")))
    ks = [int(t["n_sec"]) for t in thr]
    assert ks == sorted(ks, reverse=True) and len(thr) == len(sec_validation.THRESHOLDS)       # a higher threshold tags fewer calls


def _hip_device_count() -> int:
    import ctypes
    n = ctypes.c_int(0)
    try:
        ctypes.CDLL("libamdhip64.so").hipGetDeviceCount(ctypes.byref(n))
    except OSError:
        return 0
    return n.value


def test_filter_variants_pipeline_two_ranks(tmp_path, frozen_models):
    """Config C4 through the TOOL: `filter_variants_pipeline` under a two-process launch (RANK / WORLD_SIZE / MASTER_* as
    `python -m torch.distributed.run` sets them; torch itself is never imported) - equal-count shards, per-rank genome and
    table slices, ONE RCCL all-gather of (score, filter, flags), rank 0 writes - must produce the file the single-process
    run writes, byte for byte.  Needs two GPUs: skipped on a one-GPU box."""
    import filecmp
    import socket
    import subprocess
    import sys
    if _hip_device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL refuses two ranks on one device)")
    cs = synth.make_callset(120_000, genome_len=80_000_000, n_contigs=4, seed=31)
    d = _write_inputs(tmp_path, cs)
    common = ["--input_file", d["vcf"], "--model_file", os.path.join(GOLDEN, "synth_rf_v1.npz"), "--model_name", RF, "--runs_file", d["runs"],
              "--hpol_filter_length_dist", "10", "10", "--blacklist", d["bl"], "--reference_file", d["fa"], "--flow_order", "TGCA"] + d["ann"]
    one = str(tmp_path / "one.vcf.gz")
    subprocess.run([sys.executable, "-m", "variantcalling_amd", "filter_variants_pipeline"] + common + ["--output_file", one], check=True,
                   cwd=os.path.dirname(GOLDEN) + "/..", timeout=600)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    two = str(tmp_path / "two.vcf.gz")
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", LOCAL_WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-m", "variantcalling_amd", "filter_variants_pipeline"] + common + ["--output_file", two],
                                      env=env, cwd=os.path.dirname(GOLDEN) + "/.."))
    assert [p.wait(timeout=900) for p in procs] == [0, 0]
    assert filecmp.cmp(one, two, shallow=False)
    assert filecmp.cmp(one + ".tbi", two + ".tbi", shallow=False)
