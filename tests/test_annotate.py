"""annotate.py: feature matrix -> the reference's annotation columns (CPU, against the oracle's intermediates), and the
whole annotate_concordance call on the GPU (frame in, frame out)."""
import numpy as np
import pytest

from variantcalling_amd import annotate, schema as S, synth


def _expected_columns(ft, tracks):
    from oracle import oracle as O
    n = ft["X"].shape[0]
    exp = {"indel": ft["indel_classify"] != 0,
           "indel_classify": np.array([None, "ins", "del"], dtype=object)[ft["indel_classify"]],
           "indel_length": ft["indel_length"], "hmer_indel_length": ft["hmer_indel_length"],
           "hmer_indel_nuc": np.array([S.CODE_TO_CHAR[c] if h > 0 else None for c, h in zip(ft["hmer_indel_nuc"], ft["hmer_indel_length"])], dtype=object),
           "left_motif": np.array([O.motif_to_str(int(c)) for c in ft["left_motif"]], dtype=object),
           "right_motif": np.array([O.motif_to_str(int(c)) for c in ft["right_motif"]], dtype=object),
           "cycleskip_status": np.array(S.CSS_NAMES, dtype=object)[ft["cycleskip_status"]],
           "variant_type": np.array(S.GROUP_NAMES, dtype=object)[ft["group"]]}
    assert all(len(v) == n for v in exp.values())
    return exp


def test_columns_from_features_vocabulary():
    from oracle import oracle as O
    cs = synth.make_callset(8000, genome_len=4_000_000, n_contigs=3, seed=21)
    ft = O.featurize(cs.variants, cs.ref, cs.runs, cs.tracks)
    stems = [t.name for t in cs.tracks]
    a = annotate.columns_from_features(ft["X"], ft["group"], stems)
    assert list(a.keys()) == ["indel", "indel_classify", "indel_length", "hmer_indel_length", "hmer_indel_nuc", "left_motif", "right_motif",
                              "gc_content", "cycleskip_status", "inside_hmer_run", "close_to_hmer_run", "variant_type"] + stems
    for k, v in _expected_columns(ft, cs.tracks).items():
        assert np.array_equal(a[k], v), k
    gc = ft["X"][:, S.BASE_FEATURES.index("gc_content")]
    assert np.array_equal(a["gc_content"], np.round(gc.astype(np.float64) * 10) / 10) and set(np.unique(a["gc_content"] * 10 % 1)) == {0.0}
    assert set(a["cycleskip_status"]) <= set(S.CSS_NAMES) and (a["cycleskip_status"][a["indel"]] == "NA").all()
    assert all(len(m) == S.MOTIF_SIZE and set(m) <= set("ACGTN") for m in a["left_motif"][:500])
    names = S.feature_names(len(stems))
    for t, s in enumerate(stems):
        assert np.array_equal(a[s], ft["X"][:, names.index(f"track{t}")] > 0)
    with pytest.raises(ValueError, match="feature matrix"):
        annotate.columns_from_features(ft["X"][:, :-1], ft["group"], stems)


@pytest.mark.gpu
def test_annotate_concordance_on_the_gpu(engine):
    from oracle import oracle as O
    from variantcalling_amd.io import concordance
    cs = synth.make_callset(20_000, genome_len=8_000_000, n_contigs=3, seed=8)
    vt = cs.variants
    fr = concordance.table_to_frame(vt, cs.ref.names, (np.arange(vt.n) % 3 - 1).astype(np.int8))
    # shuffle the rows, blank every 11th call (a missed truth variant) and put one row on an unknown contig
    rng = np.random.default_rng(0)
    perm = rng.permutation(vt.n)
    fr = type(fr)([(k, v[perm]) for k, v in fr.items()])
    missing = np.arange(vt.n) % 11 == 0
    fr["alleles"][missing] = None
    fr["chrom"][5 if not missing[5] else 6] = "chrUn_x"
    out, stems = annotate.annotate_concordance(fr, engine, cs.ref, cs.runs, cs.tracks, "TGCA", (10, 10))
    assert stems == [t.name for t in cs.tracks] and out.n_rows == fr.n_rows
    assert list(out.keys())[:len(fr)] == list(fr.keys())
    vt2, rows, _ = concordance.frame_to_table(fr, cs.ref.names)
    ft = O.featurize(vt2, cs.ref, cs.runs, cs.tracks)
    exp = annotate.columns_from_features(ft["X"], ft["group"], stems)
    left = np.setdiff1d(np.arange(fr.n_rows), rows)
    assert left.size == int(missing.sum()) + 1
    for k, v in exp.items():
        got = out[k]
        assert np.array_equal(got[rows], v, equal_nan=v.dtype.kind == "f"), k
        if v.dtype == object:
            assert all(x is None for x in got[left]), k
        elif v.dtype.kind == "f":
            assert np.isnan(got[left]).all(), k
        else:
            assert not got[left].any(), k
