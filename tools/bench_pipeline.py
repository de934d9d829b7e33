"""End-to-end timing of the drop-in `filter_variants_pipeline` CLI (config C1 shape, scaled): stage seconds of the
native-codec pipeline, and the same VCF through the pure-Python codec (the per-record loops the reference runs with
pysam) for comparison.  Usage: python tools/bench_pipeline.py [n_variants]"""
import gzip
import os
import sys
import tempfile
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from variantcalling_amd import schema as S, synth  # noqa: E402
from variantcalling_amd.io import bed, fasta, vcf as pyvcf  # noqa: E402
from variantcalling_amd.pipelines import filter_variants_pipeline  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
cs = synth.make_callset(n, genome_len=400_000_000, n_contigs=4, seed=9)
with tempfile.TemporaryDirectory() as d:
    fa = os.path.join(d, "ref.fa"); fasta.write_fasta(fa, cs.ref)
    if not os.environ.get("UGVC_BENCH_NO_FAI"):
        fasta.write_fai(fa, cs.ref)          # ("Indexed reference FASTA file": the readers start together)
    vcf = os.path.join(d, "calls.vcf.gz"); pyvcf.write_vcf_from_table(vcf, cs.variants, cs.ref.names)
    runs = os.path.join(d, "runs.bed"); bed.write_bed(runs, cs.runs, cs.ref.names)
    ann = []
    for t, tr in zip(("LCR-hs38", "exome.twist", "mappability.0"), cs.tracks):
        p = os.path.join(d, f"{t}.bed"); bed.write_bed(p, tr, cs.ref.names); ann += ["--annotate_intervals", p]
    bl = os.path.join(d, "blacklist.npy"); np.save(bl, cs.blacklist)
    out = os.path.join(d, "filtered.vcf.gz")
    argv = ["filter_variants_pipeline", "--input_file", vcf, "--model_file", os.path.join(ROOT, "tests", "golden", "synth_rf_v1.npz"),
            "--model_name", "rf_model_ignore_gt_incl_hpol_runs", "--runs_file", runs, "--hpol_filter_length_dist", "10", "10",
            "--blacklist", bl, "--reference_file", fa, "--flow_order", "TGCA", "--output_file", out] + ann
    t0 = time.perf_counter()
    filter_variants_pipeline.run(argv)
    total = time.perf_counter() - t0
    st = filter_variants_pipeline.run.last_stage_seconds
    print(f"filter_variants_pipeline, {cs.variants.n} variants ({os.path.getsize(vcf) / 1e6:.1f} MB BGZF in), "
          f"{os.cpu_count()} host threads: total {total:.2f} s")
    for k, v in st.items():
        print(f"   {k:48s} {v:8.3f} s")
    from variantcalling_amd.pipelines import common
    print("   readers of the first stage (concurrent; seconds each): " + ", ".join(f"{k} {v:.3f}" for k, v in getattr(common.load_side_tables, "last_seconds", {}).items()))
    if n > 2_000_000 and len(sys.argv) < 3:
        sys.exit(0)                      # (the pure-Python comparison takes minutes at this size: pass any second argument to run it)
    t0 = time.perf_counter(); a = pyvcf.read_vcf(vcf, cs.ref.names); t_r = time.perf_counter() - t0
    res = S.FilterResult(np.zeros(a.table.n, np.float32), np.zeros(a.table.n, np.uint8), np.zeros(a.table.n, np.uint8))
    t0 = time.perf_counter(); pyvcf.write_filtered_vcf(os.path.join(d, "py.vcf.gz"), a, res); t_w = time.perf_counter() - t0
    print(f"pure-Python codec on the same file (per-record loops, as the reference's pysam code): read {t_r:.2f} s, write-back {t_w:.2f} s")
    n_out = sum(1 for ln in gzip.open(out, "rt") if not ln.startswith("#"))
    assert n_out == cs.variants.n
