"""The N > 1 orchestration of `filter_variants_pipeline` on CPU (world size 2 and 3): two / three processes rendezvous over
`dist.Group` exactly as under `python -m torch.distributed.run` (RANK / WORLD_SIZE / MASTER_*), every rank scores its
equal-count shard against ITS slice of the genome and of the side tables, the shards are reassembled in rank order, rank 0
alone writes - and the file equals the single-process run's byte for byte.  The GPU engine is replaced by the oracle
(tests/fake_gpu_driver.py: test infrastructure); the same flow on real GPUs + RCCL is
tests/test_gpu_pipelines.py::test_filter_variants_pipeline_two_ranks."""
import filecmp
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN
from variantcalling_amd import synth
from variantcalling_amd.io import bed, fasta, vcf as vcfio

RF = "rf_model_ignore_gt_incl_hpol_runs"
DRIVER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_gpu_driver.py")


def _inputs(tmp_path, cs):
    fa = str(tmp_path / "ref.fa"); fasta.write_fasta(fa, cs.ref)
    calls = str(tmp_path / "calls.vcf.gz"); vcfio.write_vcf_from_table(calls, cs.variants, cs.ref.names)
    runs = str(tmp_path / "runs.bed"); bed.write_bed(runs, cs.runs, cs.ref.names)
    ann = []
    for t, tr in zip(("LCR-hs38", "exome.twist", "mappability.0"), cs.tracks):
        p = str(tmp_path / f"{t}.bed"); bed.write_bed(p, tr, cs.ref.names); ann += ["--annotate_intervals", p]
    bl = str(tmp_path / "blacklist.npy"); np.save(bl, cs.blacklist)
    return ["filter_variants_pipeline", "--input_file", calls, "--model_file", os.path.join(GOLDEN, "synth_rf_v1.npz"), "--model_name", RF,
            "--runs_file", runs, "--hpol_filter_length_dist", "10", "10", "--blacklist", bl, "--reference_file", fa, "--flow_order", "TGCA"] + ann


@pytest.mark.parametrize("world", [2, 3])
def test_tool_shards_and_reassembles_across_ranks(tmp_path, world):
    cs = synth.make_callset(9_000, genome_len=6_000_000, n_contigs=3, seed=world)
    argv = _inputs(tmp_path, cs)
    env0 = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    one = str(tmp_path / "one.vcf.gz")
    ex1 = tmp_path / "ex1"; ex1.mkdir()
    subprocess.run([sys.executable, DRIVER, str(ex1)] + argv + ["--output_file", one], check=True, env=env0, timeout=600)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    many = str(tmp_path / "many.vcf.gz")
    exn = tmp_path / "exn"; exn.mkdir()
    procs = []
    for r in range(world):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, DRIVER, str(exn)] + argv + ["--output_file", many], env=env))
    assert [p.wait(timeout=600) for p in procs] == [0] * world
    assert sorted(os.listdir(exn)) == [f"rank{r}.npz" for r in range(world)]          # every rank scored a shard
    sizes = [int(np.load(exn / f"rank{r}.npz")["ts"].size) for r in range(world)]
    assert sum(sizes) == cs.variants.n and max(sizes) - min(sizes) <= 1               # equal-count (+-1) shards
    assert filecmp.cmp(one, many, shallow=False)
    assert filecmp.cmp(one + ".tbi", many + ".tbi", shallow=False)


def test_unsorted_multiallelic_input_across_ranks(tmp_path):
    """Round 4: ranks > 0 tokenise only their slice of the records.  That slice is a shard of the SORTED callset only for a
    sorted file: an unsorted one sends every rank back to the whole file; multi-allelic records stay with one rank (shards are
    cut between records, not between allele rows).  Either way the N-rank file equals the single-process one."""
    import gzip
    cs = synth.make_callset(4_000, genome_len=3_000_000, n_contigs=2, seed=5)
    argv = _inputs(tmp_path, cs)
    calls = argv[argv.index("--input_file") + 1]
    lines = gzip.open(calls, "rt").read().split("\n")
    hdr = [l for l in lines if l.startswith("#")]
    rec = [l.split("\t") for l in lines if l and not l.startswith("#")]
    for k in range(0, len(rec), 37):                                  # multi-allelic records, some straddling the shard seams
        f = rec[k]
        if len(f[3]) == 1 and len(f[4]) == 1:
            other = [x for x in "ACGT" if x not in (f[3], f[4])][0]
            f[4] += "," + other
            keys = f[8].split(":")
            vals = f[9].split(":")
            if "AD" in keys:
                vals[keys.index("AD")] += ",3"
            f[9] = ":".join(vals)
    for variant, name in ((rec, "sorted_ma.vcf"), (rec[:100][::-1] + rec[100:], "unsorted_ma.vcf")):
        path = str(tmp_path / name)
        open(path, "w").write("\n".join(hdr + ["\t".join(f) for f in variant]) + "\n")
        a2 = list(argv)
        a2[a2.index("--input_file") + 1] = path
        env0 = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        one = str(tmp_path / (name + ".one.vcf.gz"))
        ex1 = tmp_path / (name + ".ex1"); ex1.mkdir()
        subprocess.run([sys.executable, DRIVER, str(ex1)] + a2 + ["--output_file", one], check=True, env=env0, timeout=600)
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        many = str(tmp_path / (name + ".many.vcf.gz"))
        exn = tmp_path / (name + ".exn"); exn.mkdir()
        procs = []
        for r in range(3):
            env = dict(env0, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="3", LOCAL_WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, DRIVER, str(exn)] + a2 + ["--output_file", many], env=env))
        assert [p.wait(timeout=600) for p in procs] == [0, 0, 0]
        assert filecmp.cmp(one, many, shallow=False), name
