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
