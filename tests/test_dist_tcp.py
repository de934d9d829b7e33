"""world_size > 1 tests of the N > 1 path on CPU: equal-count sharding of a sorted callset, per-rank
scoring, padded all-gather, rank-order reassembly == scoring the whole callset.

On the GPU box the per-rank compute is libugvc_mi355x.so and the gather is RCCL (csrc/comm.hip); here
the same host plumbing (variantcalling_amd/dist.py - a plain TCP rendezvous on the launcher's
environment variables, no torch - and shard.py) runs with its host-side stand-in for the collective and
the oracle as the per-shard scorer (tests may use the oracle; the product path cannot)."""
import os
import socket
import subprocess
import sys

import numpy as np

from conftest import ROOT

WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ["UGVC_ROOT"])
from oracle import oracle as O
from variantcalling_amd import dist, model_io, shard, synth
grp = dist.Group()
W = int(os.environ["WORLD_SIZE"])
assert grp.world == W
assert "torch" not in sys.modules, "the process group must not pull torch in"
cs = synth.make_callset(3001, genome_len=2_500_000, n_contigs=3, seed=17)
forests = model_io.load_models(os.path.join(os.environ["UGVC_ROOT"], "tests", "golden", "synth_rf_v1.npz"))[
    "rf_model_ignore_gt_incl_hpol_runs"]
mine = shard.shard_of(cs.variants, grp.rank, grp.world)
local = O.filter_variants(mine, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
full = grp.allgather_results_host(local, cs.variants.n)
uid = grp.broadcast_bytes(b"x" * 128 if grp.rank == 0 else None, 0)
assert uid == b"x" * 128
assert grp.max_float(float(grp.rank)) == float(W - 1) and grp.sum_float(1.0) == float(W)
assert grp.broadcast_bytes(b"from-last" if grp.rank == W - 1 else None, W - 1) == b"from-last"
grp.barrier()
np.savez(os.environ["UGVC_OUT"] + f".{grp.rank}.npz", score=full.tree_score, filter=full.filter, flags=full.flags,
         n_local=mine.n)
grp.close()
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _check(tmp_path, frozen_models, world):
    from oracle import oracle as O
    from variantcalling_amd import shard, synth
    cs = synth.make_callset(3001, genome_len=2_500_000, n_contigs=3, seed=17)
    exp = O.filter_variants(cs.variants, cs.ref, cs.runs, cs.tracks, cs.blacklist,
                            frozen_models["rf_model_ignore_gt_incl_hpol_runs"])
    sizes = []
    for rank in range(world):
        z = np.load(str(tmp_path / "out") + f".{rank}.npz")
        assert np.array_equal(z["score"], exp.tree_score), rank
        assert np.array_equal(z["filter"], exp.filter) and np.array_equal(z["flags"], exp.flags), rank
        sizes.append(int(z["n_local"]))
    b = shard.shard_bounds(cs.variants.n, world)
    assert sizes == [int(b[r + 1] - b[r]) for r in range(world)] and sum(sizes) == cs.variants.n


def test_two_rank_shard_gather_under_the_torchrun_launcher(tmp_path, frozen_models):
    """Launched exactly as the driver launches bench.py: `python -m torch.distributed.run` sets RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT (and keeps its own store on MASTER_PORT); the workers rendezvous beside it without torch."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, UGVC_ROOT=ROOT, UGVC_OUT=str(tmp_path / "out"), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    _check(tmp_path, frozen_models, 2)


def test_three_ranks_from_plain_environment_variables(tmp_path, frozen_models):
    """Any launcher that sets the five variables will do: three bare subprocesses, odd world size."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(3):
        env = dict(os.environ, UGVC_ROOT=ROOT, UGVC_OUT=str(tmp_path / "out"), OMP_NUM_THREADS="1", RANK=str(rank),
                   LOCAL_RANK=str(rank), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-3000:]
    _check(tmp_path, frozen_models, 3)


def test_eight_ranks_under_the_torchrun_launcher(tmp_path, frozen_models):
    """The world size the node has GPUs for: eight workers behind `torch.distributed.run`, rank 0 serving the star, every
    rank ending up with the whole callset in order."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, UGVC_ROOT=ROOT, UGVC_OUT=str(tmp_path / "out"), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    _check(tmp_path, frozen_models, 8)
