"""TEST INFRASTRUCTURE: runs `filter_variants_pipeline.run(argv)` with the GPU engine replaced by a stand-in that scores
with the CPU oracle and "all-gathers" through files - so the tool's multi-rank ORCHESTRATION (equal-count shards, per-rank
context slices, rank-order reassembly, multi-allelic fold, rank-0-only write, rendezvous over dist.Group) runs in this
GPU-less container with WORLD_SIZE = 2.  The product path never imports this module.
Usage: python tests/fake_gpu_driver.py <exchange dir> <tool argv...>"""
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from variantcalling_amd import engine as real_engine, schema as S  # noqa: E402

EXCHANGE = sys.argv[1]


class FakeEngine:
    def __init__(self, device=0):
        self.device = device
        self.rank, self.world = 0, 1
        self.cfg = None
        self.vt = None
        self.res = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def reserve(self, n_variants, alleles_len):               # (Engine.reserve: allocations only)
        assert n_variants >= 0 and alleles_len >= 0

    def filter_variants(self, vt):
        ref, runs, tracks, bl, forests, flow, hp_len, hp_dist, mark = self.cfg
        return O.filter_variants(vt, ref, runs, tracks, bl, forests, flow, hp_len, hp_dist, mark)

    def upload_variants(self, vt):
        self.vt = vt

    def filter_resident(self):
        self.res = self.filter_variants(self.vt)

    def comm_unique_id(self):
        return bytes(range(128))

    def comm_init(self, uid, rank, world):
        assert uid == bytes(range(128))
        self.rank, self.world = rank, world

    def comm_info(self):
        return dict(nranks=self.world, rank=self.rank, device=self.device)

    def allgather_resident(self, cap):
        assert cap >= self.vt.n
        tmp = os.path.join(EXCHANGE, f"rank{self.rank}.tmp.npz")
        np.savez(tmp, ts=self.res.tree_score, fl=self.res.filter, fg=self.res.flags)
        os.replace(tmp, os.path.join(EXCHANGE, f"rank{self.rank}.npz"))

    def gathered_download(self, cap, world, counts):
        parts = []
        for r in range(world):
            p = os.path.join(EXCHANGE, f"rank{r}.npz")
            t0 = time.time()
            while not os.path.exists(p):
                if time.time() - t0 > 120:
                    raise TimeoutError(p)
                time.sleep(0.05)
            z = np.load(p)
            assert z["ts"].size == counts[r]
            parts.append(z)
        return S.FilterResult(np.concatenate([z["ts"] for z in parts]), np.concatenate([z["fl"] for z in parts]),
                              np.concatenate([z["fg"] for z in parts]))


def fake_configure(eng, ref, runs, tracks, bl, forests, flow_order="TGCA", hpol_len=10, hpol_dist=10, mark_hpol=True):
    eng.cfg = (ref, runs, tracks, bl, forests, flow_order, hpol_len, hpol_dist, mark_hpol)
    return eng


real_engine.Engine = FakeEngine
real_engine.configure = fake_configure
from variantcalling_amd.pipelines import filter_variants_pipeline  # noqa: E402

sys.exit(filter_variants_pipeline.run(sys.argv[2:]) or 0)
