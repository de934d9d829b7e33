"""bench.py's own N > 1 bookkeeping on CPU (world size 2 and 3): the processes rendezvous over `dist.Group` as under
`python -m torch.distributed.run`, the SAME callset is cut into equal-count shards, every rank configures ITS slice of the genome
and of the side tables, the steps end with the (stand-in) all-gather, the clock is the max over ranks, and rank 0 prints one JSON
line.  The engine is the CPU oracle (tests/fake_gpu_driver.py --bench: test infrastructure), so what is tested is the part the
driver's SCALE run depends on and no single-GPU run exercises: `n_gpus`, `variants_per_gpu`, `rccl_nranks`, `gather_consistent`,
the every-row comparison of the GATHERED callset with the oracle on the unsharded tables (--check-rows -1), the value's arithmetic."""
import json
import os
import socket
import subprocess
import sys

import pytest

DRIVER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_gpu_driver.py")
N_VAR = 30_000


def _run(tmp_path, world, extra=()):
    env0 = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ex = tmp_path / f"ex{world}"
    ex.mkdir()
    argv = ["--bench", "--gpus", str(world), "--variants", str(N_VAR), "--steps", "2", "--warmup", "1", "--spinup", "0", "--cpu-sample", "0",
            "--check-rows", "-1", "--no-e2e"] + list(extra)
    procs = []
    for r in range(world):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, DRIVER, str(ex)] + argv, env=env, stdout=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert [p.returncode for p in procs] == [0] * world
    lines = [[l for l in o.splitlines() if l.startswith("{")] for o in outs]
    assert [len(l) for l in lines] == [1] + [0] * (world - 1)              # rank 0 alone prints, and exactly one line
    return json.loads(lines[0][0]), ex


@pytest.mark.parametrize("world", [2, 3])
def test_bench_shards_gathers_and_reports(tmp_path, world):
    import numpy as np
    d, ex = _run(tmp_path, world)
    assert d["n_gpus"] == world and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "strong" and d["higher_is_better"] is True
    cfg = d["config"]
    n_total = int(cfg["workload"].split()[1])
    assert abs(n_total - N_VAR) < 100 and cfg["rccl_nranks"] == world and cfg["sharding"].startswith(f"equal-count x{world}")
    sizes = [int(np.load(ex / f"rank{r}.npz")["ts"].size) for r in range(world)]
    # equal-count shards; a cut may have moved to a contig change within 1 % of the share (shard.shard_bounds with the contig column)
    assert sum(sizes) == n_total and max(sizes) <= n_total / world * 1.01 + 1 and cfg["variants_per_gpu"] == sizes[0]
    par = d["parity"]
    # rank 0's shard, scored against ITS slice of the tables, equals the oracle on the whole tables; every rank found its shard in
    # the gathered columns; the gathered callset equals the oracle row for row
    assert par["oracle_slice_bit_exact"] is True and par["oracle_rows_checked"] == sizes[0]
    assert par["gather_consistent"] is True and par["gathered_all_rows_bit_exact"] is True
    # whole-job throughput: all ranks' variants over the max-over-ranks wall clock of the K steps
    assert d["value"] == pytest.approx(n_total * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"]), rel=1e-9)
    assert d["roofline"]["variants_per_launch"] == sizes[0] and d["cpu_baseline"] is None and d["e2e_incl_pcie"] is None


def test_bench_refuses_a_world_that_is_not_the_gpus_flag(tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, DRIVER, str(tmp_path), "--bench", "--gpus", "2", "--variants", "2000", "--steps", "1", "--warmup", "0",
                        "--spinup", "0", "--cpu-sample", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr
