"""CPU tests of the host-side I/O adapters and of the evaluation maths (no GPU)."""
import gzip
import os
import pickle

import numpy as np
import pytest

from conftest import GOLDEN
from variantcalling_amd import evaluate, schema as S, synth
from variantcalling_amd.io import bed, fasta, vcf as vcfio


@pytest.fixture(scope="module")
def cs():
    return synth.make_callset(3000, genome_len=2_000_000, n_contigs=3, seed=41)


def test_fasta_round_trip(tmp_path, cs):
    for name in ("r.fa", "r.fa.gz"):
        p = str(tmp_path / name)
        fasta.write_fasta(p, cs.ref, width=61)
        back = fasta.read_fasta(p)
        assert back.names == cs.ref.names and np.array_equal(back.contig_off, cs.ref.contig_off)
        assert np.array_equal(back.codes, cs.ref.codes)
    p = str(tmp_path / "mixed.fa")
    open(p, "w").write(">c1 desc\nACGTNacgtn\nRYK\n>c2\n\nGG\r\nTT\n")
    r = fasta.read_fasta(p)
    assert r.names == ["c1", "c2"] and r.contig_off.tolist() == [0, 13, 17]
    assert S.decode_bases(r.codes) == "ACGTNACGTNNNN" + "GGTT"
    assert fasta.read_fasta(p, contigs=["c2"]).names == ["c2"]
    with pytest.raises(ValueError):
        open(p, "w").write("ACGT\n")
        fasta.read_fasta(p)


def test_bed_merge_and_interval_list(tmp_path):
    names = ["chr1", "chr2"]
    p = str(tmp_path / "LCR-hs38.bed")
    open(p, "w").write("track name=x\n#c\nchr2\t5\t9\nchr1\t100\t200\nchr1\t150\t180\nchr1\t190\t260\nchr1\t260\t300\n"
                       "chr1\t10\t20\nchrUn\t1\t5\nchr1\t400\t400\n")
    t = bed.read_intervals(p, names)
    assert t.name == "LCR-hs38"
    assert t.starts.tolist() == [10, 100, 5] and t.ends.tolist() == [20, 300, 9]     # book-ended intervals merge too
    assert t.contig_ptr.tolist() == [0, 2, 3]
    raw = bed.read_intervals(p, names, merge=False)
    assert raw.starts.size == 6
    q = str(tmp_path / "runs.interval_list")
    open(q, "w").write("@HD\tVN:1.6\n@SQ\tSN:chr1\tLN:1000\nchr1\t11\t20\t+\tx\n")
    t2 = bed.read_intervals(q, names)
    assert t2.starts.tolist() == [10] and t2.ends.tolist() == [20]
    # merged tables always satisfy the engine's precondition
    rng = np.random.default_rng(0)
    st = rng.integers(0, 10_000, 2000)
    tr = bed.track_from_arrays(rng.integers(0, 2, 2000), st, st + rng.integers(1, 500, 2000), 2)
    for c in range(2):
        s, e = tr.starts[tr.contig_ptr[c]:tr.contig_ptr[c + 1]], tr.ends[tr.contig_ptr[c]:tr.contig_ptr[c + 1]]
        assert (np.diff(s) > 0).all() and (np.diff(e) > 0).all() and (e[:-1] < s[1:]).all()
    bed.write_bed(str(tmp_path / "o.bed"), tr, names)
    again = bed.read_intervals(str(tmp_path / "o.bed"), names)
    assert np.array_equal(again.starts, tr.starts) and np.array_equal(again.ends, tr.ends)



def test_track_from_arrays_unsorted_unsigned_contigs():
    """a decreasing contig column of an UNSIGNED dtype must not pass the sortedness shortcut (differences wrap)"""
    from variantcalling_amd.io import bed
    contig = np.array([1, 1, 0, 0], np.uint16)
    starts, ends = np.array([10, 50, 5, 30]), np.array([20, 60, 9, 40])
    tr = bed.track_from_arrays(contig, starts, ends, 2, merge=False)
    ref = bed.track_from_arrays(contig.astype(np.int64), starts, ends, 2, merge=False)
    assert tr.starts.tolist() == ref.starts.tolist() == [5, 30, 10, 50] and tr.contig_ptr.tolist() == [0, 2, 4]
    assert tr.ends.tolist() == [9, 40, 20, 60]

def test_blacklist_formats(tmp_path):
    names = ["chr1", "chr2"]
    want = np.array([(0 << 32) | 5, (0 << 32) | 9, (1 << 32) | 7], dtype=np.uint64)
    p = str(tmp_path / "b.pkl")
    pickle.dump([("chr1", 9), ("chr2", 7), ("chr1", 5), ("chrUn", 3)], open(p, "wb"))
    assert np.array_equal(bed.read_blacklist(p, names), want)
    pickle.dump({"a": {("chr1", 5), ("chr1", 9)}, "b": [("chr2", 7)]}, open(p, "wb"))
    assert np.array_equal(bed.read_blacklist(p, names), want)
    import pandas as pd
    pickle.dump(pd.DataFrame({"chrom": ["chr1", "chr2", "chr1"], "pos": [9, 7, 5]}), open(p, "wb"))
    assert np.array_equal(bed.read_blacklist(p, names), want)
    q = str(tmp_path / "b.bed")
    open(q, "w").write("chr1\t4\t5\nchr1\t8\t9\nchr2\t6\t7\n")
    assert np.array_equal(bed.read_blacklist(q, names), want)
    np.save(str(tmp_path / "k.npy"), want[::-1])
    assert np.array_equal(bed.read_blacklist(str(tmp_path / "k.npy"), names), want)
    # --blacklist cohort_fp.h5: loci as a (chrom, pos) MultiIndex or as columns of a pandas fixed-format frame
    from variantcalling_amd.io import h5
    chrom, pos = np.array(["chr1", "chr2", "chr1", "chrUn"], dtype=object), np.array([9, 7, 5, 3])
    h5.write_hdf(str(tmp_path / "x.h5"), {"blacklist": h5.Frame([("n", np.arange(4))], index=[chrom, pos], index_names=["chrom", "pos"])})
    assert np.array_equal(bed.read_blacklist(str(tmp_path / "x.h5"), names), want)
    h5.write_hdf(str(tmp_path / "y.hdf"), {"a": h5.Frame([("chrom", chrom[:2]), ("pos", pos[:2])]),
                                            "b": h5.Frame([("chrom", chrom[2:]), ("pos", pos[2:])]), "c": h5.Frame([("z", np.zeros(1))])})
    assert np.array_equal(bed.read_blacklist(str(tmp_path / "y.hdf"), names), want)
    h5.write_hdf(str(tmp_path / "z.h5"), {"c": h5.Frame([("z", np.zeros(1))])})
    with pytest.raises(ValueError, match="no frame or series"):
        bed.read_blacklist(str(tmp_path / "z.h5"), names)
    with pytest.raises(ValueError, match="unsupported blacklist format"):
        bed.read_blacklist(str(tmp_path / "x.txt"), names)


def test_vcf_round_trip_and_write_back(tmp_path, cs):
    vt = cs.variants
    ids = np.arange(vt.n) % 3 == 0
    for name in ("in.vcf", "in.vcf.gz"):
        p = str(tmp_path / name)
        vcfio.write_vcf_from_table(p, vt, cs.ref.names, ids=ids)
        v = vcfio.read_vcf(p, cs.ref.names)
        for c in S.VariantTable.COLS:
            assert np.array_equal(getattr(v.table, c), getattr(vt, c)), c
        assert np.array_equal(v.table.alleles, vt.alleles) and np.array_equal(v.ids, ids)
    rng = np.random.default_rng(1)
    res = S.FilterResult(rng.random(vt.n).astype(np.float32), rng.integers(0, 2, vt.n).astype(np.uint8),
                         rng.integers(0, 4, vt.n).astype(np.uint8))
    out = str(tmp_path / "out.vcf.gz")
    vcfio.write_filtered_vcf(out, v, res)
    raw = open(out, "rb").read()
    assert raw.endswith(vcfio._BGZF_EOF) and raw[:4] == b"\x1f\x8b\x08\x04" and raw[12:14] == b"BC"
    lines = gzip.open(out, "rt").read().splitlines()
    hdr = [x for x in lines if x.startswith("#")]
    assert any(x.startswith("##FILTER=<ID=LOW_SCORE") for x in hdr) and any("ID=TREE_SCORE" in x for x in hdr)
    assert hdr[-1].startswith("#CHROM")
    recs = [x.split("\t") for x in lines if not x.startswith("#")]
    assert len(recs) == vt.n
    for k in (0, 1, 17, vt.n - 1):
        f = recs[k]
        tags = set(f[6].split(";"))
        assert ("LOW_SCORE" in tags) == (res.filter[k] == 1)
        assert ("HPOL_RUN" in tags) == bool(res.flags[k] & 1) and ("COHORT_FP" in tags) == bool(res.flags[k] & 2)
        assert (tags == {"PASS"}) == (res.filter[k] == 0 and res.flags[k] & 3 == 0)
        info = dict(x.split("=") if "=" in x else (x, True) for x in f[7].split(";"))
        assert np.float32(info["TREE_SCORE"]) == res.tree_score[k] and "SOR" in info
        assert ("HPOL_RUN" in info) == bool(res.flags[k] & 1)


def test_vcf_edge_records(tmp_path):
    p = str(tmp_path / "e.vcf")
    open(p, "w").write(
        "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ts1\ts2\n"
        "chr2\t50\t.\tA\tG,T\t.\t.\tDP=3;SOR=1.5;TLOD=4.5,7.25\tGT:AD:DP:GQ\t1|1:0,7,1:8:99\t0/0:1,0,0:1:3\n"
        "chr1\t10\trs5\tAT\tA\t33.5\tLowQual\t.\tGT:DP\t./.:.\t0/1:2\n"
        "chr1\t10\t.\tn\t<DEL>\t7\tPASS\tSOR=.\n")
    v = vcfio.read_vcf(p, ["chr1", "chr2"])
    t = v.table
    assert t.contig.tolist() == [0, 0, 1] and t.pos.tolist() == [10, 10, 50] and v.order.tolist() == [1, 2, 0]
    assert t.qual.tolist() == [33.5, 7.0, 0.0] and t.sor.tolist() == [0.0, 0.0, 1.5]
    assert t.dp.tolist() == [0, 0, 8] and t.ad_alt.tolist() == [0, 0, 7] and t.gq.tolist() == [0, 0, 99]
    assert t.gt.tolist() == [0, 0, 2] and v.ids.tolist() == [True, False, False]
    assert S.decode_bases(t.alleles[t.alt_off[2]: t.alt_off[2] + t.alt_len[2]]) == "G"     # first ALT
    assert t.alt_len[1] == 5 and (t.alleles[t.alt_off[1]: t.alt_off[1] + 5] == 0).all()     # symbolic -> N
    m = vcfio.read_vcf(p, ["chr1", "chr2"], is_mutect=True)
    assert m.table.qual.tolist() == [0.0, 0.0, 72.5]
    out = str(tmp_path / "o.vcf")
    vcfio.write_filtered_vcf(out, v, S.FilterResult(np.array([.5, .25, 1], np.float32), np.array([1, 0, 0], np.uint8),
                                                    np.array([0, 1, 2], np.uint8)))
    recs = [x.split("\t") for x in open(out).read().splitlines() if not x.startswith("#")]
    assert [r[0] for r in recs] == ["chr2", "chr1", "chr1"]                     # input order kept
    assert recs[0][6] == "COHORT_FP" and recs[1][6] == "LOW_SCORE" and recs[2][6] == "HPOL_RUN"
    assert recs[1][7] == "TREE_SCORE=0.5" and recs[2][7] == "SOR=.;TREE_SCORE=0.25;HPOL_RUN"
    with pytest.raises(ValueError, match="not in the reference"):
        vcfio.read_vcf(p, ["chr1"])


# ------------------------------------------------------------------ evaluation maths vs the reference run here
def test_precision_recall_curve_matches_reference_run():
    z = np.load(os.path.join(GOLDEN, "reference_run_v1.npz"))
    for case in range(4):
        p, r, f1, thr = evaluate.precision_recall_curve(z[f"pr{case}_labels"], z[f"pr{case}_scores"], z[f"pr{case}_fn"],
                                                        pos_label=1, min_class_counts_to_output=int(z[f"pr{case}_min_cls"]))
        assert np.array_equal(p, z[f"pr{case}_precision"]) and np.array_equal(r, z[f"pr{case}_recall"]), case
        assert np.array_equal(f1, z[f"pr{case}_f1"]) and np.array_equal(thr, z[f"pr{case}_thr"]), case
    assert np.array_equal(evaluate.get_precision(z["prec_fp"], z["prec_tp"]), z["prec_out"])
    assert np.array_equal(evaluate.get_recall(z["prec_fp"], z["prec_tp"]), z["rec_out"])
    assert np.allclose(evaluate.get_f1(z["prec_out"], z["rec_out"]), z["f1_out"], rtol=1e-15, atol=0)
    # reference KATs (test/unit/utils/test_stats_utils.py:112-157)
    assert np.isclose(evaluate.get_precision(100, 900), 0.9) and round(float(evaluate.get_precision(1, 900)), 5) == 0.99889
    assert round(float(evaluate.get_f1(0.9, 0.99)), 6) == 0.942857 and round(float(evaluate.get_f1(0.9, 0.5)), 6) == 0.642857
    lab = np.array([0, 1] * 50 + [1] * 10); sc = np.array([0.1, 0.8] * 50 + [-1] * 10, float)
    p, r, f1, _ = evaluate.precision_recall_curve(lab, sc, np.r_[np.zeros(100, bool), np.ones(10, bool)], 1, 1)
    assert p.size == 1 and round(float(f1.max()), 9) == 0.909090909
    assert evaluate.precision_recall_curve([], [], np.array([]), 1, 1)[0].size == 0


def test_calc_performance_and_accuracy_table():
    rng = np.random.default_rng(2)
    n = 4000
    tp = rng.random(n) < 0.6
    fn = ~tp & (rng.random(n) < 0.2)
    fp = ~tp & ~fn
    score = np.clip(rng.normal(0.35 + 0.4 * tp, 0.15), 0, 1)
    passed = score > 0.5
    res, curve = evaluate.calc_performance(score, passed, tp, fp, fn, missing_candidate=fn & (rng.random(n) < 0.5))
    assert res["tp"] == int((tp & passed).sum()) and res["fp"] == int((fp & passed).sum())
    assert res["fn"] == int(fn.sum() + (tp & ~passed).sum()) and res["# pos"] == int(tp.sum() + fn.sum())
    s, rec, prec, f1 = curve
    assert (np.diff(s) >= 0).all() and (np.diff(rec) <= 1e-12).all() and 0.5 < np.nanmax(f1) <= 1
    rows = evaluate.accuracy_table(score, passed, tp, rng.random(n) < 0.2, rng.integers(0, 16, n))
    assert [r["group"] for r in rows] == list(evaluate.CATEGORIES)
    snp = rows[0]
    assert snp["tp"] + snp["fn"] == snp["initial_tp"] and snp["fp"] <= snp["initial_fp"]
    assert rows[7]["initial_tp"] == sum(r["initial_tp"] for r in rows[1:7])
    # the same rows from integer counts (what Engine.eval_counts returns) and the category bit layout
    indel, hm, lab = rng.random(n) < 0.2, rng.integers(0, 16, n), rng.random(n) < 0.5
    bits = evaluate.category_bits(indel, hm)
    assert bits.dtype == np.uint16 and (bits >> len(evaluate.CATEGORIES) == 0).all()
    cnt = [[int((m & lab).sum()), int((m & ~lab).sum()), int((m & lab & passed).sum()), int((m & ~lab & passed).sum())]
           for m in (((bits >> c) & 1).astype(bool) for c in range(len(evaluate.CATEGORIES)))]
    assert evaluate.accuracy_rows(cnt) == evaluate.accuracy_table(score, passed, lab, indel, hm)


def test_pipelines_help_and_fail_loudly_without_gpu(tmp_path, capsys, cs):
    import torch
    from variantcalling_amd.pipelines import MODULES, filter_variants_pipeline, train_models_pipeline
    assert [m.__name__.split(".")[-1] for m in MODULES] == ["filter_variants_pipeline", "train_models_pipeline"]
    for mod in MODULES:
        assert mod.run.__doc__
        with pytest.raises(SystemExit) as e:
            mod.run([mod.__name__, "-h"])
        assert e.value.code == 0
    out = capsys.readouterr().out
    for flag in ("--hpol_filter_length_dist", "--blacklist_cg_insertions", "--is_mutect", "--annotate_intervals",
                 "--output_file_prefix", "--evaluate_concordance_contig", "--exome_weight_annotation", "--ignore_filter_status"):
        assert flag in out
    if torch.cuda.is_available():
        return
    fa, vc, rb = str(tmp_path / "r.fa"), str(tmp_path / "v.vcf"), str(tmp_path / "runs.bed")
    fasta.write_fasta(fa, cs.ref)
    vcfio.write_vcf_from_table(vc, cs.variants.slice(0, 50), cs.ref.names)
    bed.write_bed(rb, cs.runs, cs.ref.names)
    ann = []
    for t, tr in enumerate(cs.tracks):
        ann += ["--annotate_intervals", str(tmp_path / f"t{t}.bed")]
        bed.write_bed(ann[-1], tr, cs.ref.names)
    base = ["x", "--input_file", vc, "--model_file", os.path.join(GOLDEN, "synth_rf_v1.npz"), "--model_name",
            "rf_model_ignore_gt_incl_hpol_runs", "--runs_file", rb, "--reference_file", fa, "--output_file", str(tmp_path / "o.vcf")]
    with pytest.raises(ValueError, match="annotation track"):
        filter_variants_pipeline.run(base)
    with pytest.raises(RuntimeError):          # no CPU fallback
        filter_variants_pipeline.run(base + ann)
    with pytest.raises(RuntimeError):
        train_models_pipeline.run(["x", "--input_file", vc, "--reference", fa, "--output_file_prefix", str(tmp_path / "m")])


def test_threshold_model_is_monotone_and_scores_like_its_cells():
    """model_io.make_threshold_model (the documented two-feature "simple model"): evaluated by the oracle's tree walker it
    returns the monotone closure of the per-cell true-call fractions; never falls with QUAL, never rises with SOR."""
    from oracle import oracle as O
    from variantcalling_amd import model_io
    rng = np.random.default_rng(8)
    n = 30_000
    qual = (rng.random(n) * 60).astype(np.float32)
    sor = (rng.random(n) * 4).astype(np.float32)
    y = (qual / 60 - sor / 8 + rng.normal(0, 0.2, n)) > 0.25
    w = rng.integers(1, 4, n).astype(np.float64)
    F = 20
    f = model_io.make_threshold_model(qual, sor, y, w, i_a=0, i_b=1, n_features=F, k=16)
    assert f.kind == S.MODEL_RF and f.n_trees == 1 and f.max_depth == 8 and f.leaf_value.shape == (256, 2)
    assert np.allclose(f.leaf_value.sum(axis=1), 1.0)
    X = np.zeros((4000, F), np.float32)
    X[:, 0], X[:, 1] = qual[:4000], sor[:4000]
    score = O.forest_predict(f, X)[1]
    assert ((score > 0.5) == y[:4000]).mean() > 0.8
    # recompute the cells independently
    ca = np.unique(np.quantile(qual, np.linspace(0, 1, 17)[1:-1], method="inverted_cdf")).astype(np.float32)
    cb = np.unique(np.quantile(sor, np.linspace(0, 1, 17)[1:-1], method="inverted_cdf")).astype(np.float32)
    ia, ib = np.searchsorted(ca, qual, side="left"), np.searchsorted(cb, sor, side="left")
    tp, tot = np.zeros((16, 16)), np.zeros((16, 16))
    np.add.at(tp, (ia, ib), w * y)
    np.add.at(tot, (ia, ib), w)
    cell = (tp + 1) / (tot + 2)
    closed = np.array([[cell[:i + 1, j:].max() for j in range(16)] for i in range(16)])
    assert np.allclose(score, closed[ia[:4000], ib[:4000]])
    for fixed in (0.5, 2.0, 3.5):
        g = np.zeros((200, F), np.float32)
        g[:, 0], g[:, 1] = np.linspace(0, 60, 200), fixed
        assert np.all(np.diff(O.forest_predict(f, g)[1]) >= 0)
        g[:, 0], g[:, 1] = 30.0, np.linspace(0, 4, 200)
        assert np.all(np.diff(O.forest_predict(f, g)[1]) <= 0)


def test_fai_index_points_at_every_contigs_first_base(tmp_path):
    """io.fasta.write_fai (round 4): the samtools index of a FASTA written by write_fasta - with it beside the reference the
    tools know the contig names at once and start every side-table reader together with the FASTA reader."""
    from variantcalling_amd import synth
    from variantcalling_amd.io import fasta, vcf_native as nv
    cs = synth.make_callset(500, genome_len=1_000_003, n_contigs=3, seed=1)
    p = str(tmp_path / "r.fa")
    fasta.write_fasta(p, cs.ref)
    fasta.write_fai(p, cs.ref)
    raw = open(p, "rb").read()
    for c, line in enumerate(open(p + ".fai")):
        name, n, off, lb, lw = line.rstrip("\n").split("\t")
        assert name == cs.ref.names[c] and int(n) == cs.ref.contig_len(c) and (int(lb), int(lw)) == (60, 61)
        assert raw[int(off) - 1:int(off)] == b"\n"
        assert raw[int(off):int(off) + 1] == fasta.CODE_TO_CHAR[cs.ref.codes[cs.ref.contig_off[c]]].encode()
    assert nv.read_fasta_names(p) == list(cs.ref.names)


def test_hist_gradient_boosting_flattens_to_the_xgboost_style_table():
    """train_models_pipeline's `xgb_model_*` (round 6): a scikit-learn HistGradientBoostingClassifier stored in XGBoost's format -
    `x <= t64` of scikit-learn becomes `x < nextafter(f32_floor(t64))`, so every finite f32 input takes the same path (also rows
    sitting exactly ON a threshold), the f32 margin equals scikit-learn's f64 decision function to ~1e-5, depth <= 6 and 100 trees
    (the shape the leaf-matrix GEMM takes), and it round-trips through the .npz the filter tool loads."""
    from sklearn.ensemble import HistGradientBoostingClassifier
    from oracle import oracle as O
    from variantcalling_amd import model_io, schema as S
    rng = np.random.default_rng(0)
    X = rng.normal(size=(6000, 20)).astype(np.float32)
    X[:, 2] = rng.integers(0, 50, 6000)
    y = (X[:, 0] + X[:, 3] * X[:, 5] + 0.1 * X[:, 2] > 2.5).astype(int)
    m = HistGradientBoostingClassifier(max_iter=100, max_depth=6, max_leaf_nodes=64, learning_rate=0.1, early_stopping=False, random_state=0).fit(X, y)
    f = model_io.flatten_sklearn(m)
    assert f.kind == S.MODEL_GBT and f.n_trees == 100 and f.max_depth <= 6 and f.n_features == 20
    margin, score = O.forest_predict(f, X)
    ref = m.decision_function(X)
    assert np.abs(margin - ref).max() < 2e-5 and np.array_equal(margin > 0, ref > 0)
    assert np.abs(score - m.predict_proba(X)[:, 1]).max() < 1e-5
    # rows placed exactly on thresholds, and one float below
    inner = np.flatnonzero(f.feature >= 0)
    Xa = X[:1500].copy()
    for k in range(Xa.shape[0]):
        j = int(inner[rng.integers(0, inner.size)])
        Xa[k, f.feature[j]] = np.nextafter(f.threshold[j], np.float32(-np.inf)) if k % 2 else f.threshold[j]
    ma, _ = O.forest_predict(f, Xa)
    assert np.abs(ma - m.decision_function(Xa)).max() < 2e-5
    with pytest.raises(ValueError, match="binary"):
        model_io.flatten_hist_gbt(HistGradientBoostingClassifier(max_iter=3).fit(X[:300], rng.integers(0, 3, 300)))
