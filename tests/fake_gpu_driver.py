"""TEST INFRASTRUCTURE: runs `filter_variants_pipeline.run(argv)` - or, with `--bench`, `bench.py`'s own main() - with the GPU
engine replaced by a stand-in that scores with the CPU oracle and "all-gathers" through files, so the multi-rank ORCHESTRATION
(equal-count shards, per-rank context slices, rank-order reassembly, multi-allelic fold, rank-0-only write, rendezvous over
dist.Group; bench.py: shard bookkeeping, the max-over-ranks clock, the gather consistency and every-row checks, the JSON line)
runs in this GPU-less container with WORLD_SIZE = 2.  The product path never imports this module; the numbers such a bench
line carries are meaningless and say so (device name).
Usage: python tests/fake_gpu_driver.py <exchange dir> <tool argv...>  |  <exchange dir> --bench <bench.py argv...>"""
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

    # ---- the per-table setters of the single-process tool (round 6: a table goes up the moment its reader has it, on the
    # context's thread - filter_variants_pipeline.run); together they build what `configure` sets in one go
    def _part(self):
        if not isinstance(self.cfg, dict):
            self.cfg = dict(ref=None, runs=None, tracks={}, n_tracks=0, bl=None, forests=None, flow="TGCA", hp=(10, 10, True))
        return self.cfg

    def set_reference(self, ref):
        self._part()["ref"] = ref

    def set_runs(self, runs, min_len=10, max_dist=10, mark_hpol=True):
        c = self._part()
        assert c["ref"] is not None, "the engine checks interval tables against the reference's contigs: reference first"
        c["runs"], c["hp"] = runs, (min_len, max_dist, mark_hpol)

    def set_track(self, t, tr):
        c = self._part()
        assert c["ref"] is not None and 0 <= t < S.MAX_TRACKS
        c["tracks"][t] = tr

    def set_n_tracks(self, n):
        self._part()["n_tracks"] = n

    def set_blacklist(self, keys):
        self._part()["bl"] = keys

    def set_flow_order(self, flow):
        self._part()["flow"] = flow

    def set_models(self, forests):
        self._part()["forests"] = forests

    def _cfg_tuple(self):
        if isinstance(self.cfg, dict):
            c = self.cfg
            assert sorted(c["tracks"]) == list(range(c["n_tracks"])), "every track slot below n_tracks must have been uploaded"
            runs = c["runs"] if c["runs"] is not None and c["runs"].starts.size else None
            return (c["ref"], runs, [c["tracks"][t] for t in range(c["n_tracks"])], c["bl"], c["forests"], c["flow"]) + tuple(c["hp"])
        return self.cfg

    # ---- what bench.py asks of an engine beyond the tool's calls
    def close(self):
        pass

    def device_info(self):
        return dict(name="CPU oracle stand-in (tests/fake_gpu_driver.py): timings meaningless", n_cus=1, hbm_bytes=0)

    def device_attr(self, what):
        return {"clock_khz": 1_000_000, "n_cus": 1, "mem_clock_khz": 1, "lds_bytes": 0}[what]

    def set_kernel_variant(self, v):
        assert v == 0

    def device_sync(self):
        pass

    def selftest(self, n=1024):
        pass

    def pass_clock_ghz(self, passes=40):
        return 1.0, 0.0

    def timed_steps(self, iters, cap=0, gather=False, per_step_events=True):
        t0 = time.perf_counter()
        for _ in range(iters):
            if self.res is None:                              # (one oracle pass stands for every step: the result cannot differ)
                self.filter_resident()
            if gather:
                self.allgather_resident(cap)
        ms = (time.perf_counter() - t0) * 1e3
        self._step_ms = [ms / iters] * iters
        return ms, ms

    def last_step_ms(self, n):
        return np.asarray(self._step_ms[:n], np.float64)

    def download_results(self):
        return self.res

    def filter_variants(self, vt):
        ref, runs, tracks, bl, forests, flow, hp_len, hp_dist, mark = self._cfg_tuple()
        return O.filter_variants(vt, ref, runs, tracks, bl, forests, flow, hp_len, hp_dist, mark)

    def upload_variants(self, vt):
        self.vt = vt
        self.res = None

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
if len(sys.argv) > 2 and sys.argv[2] == "--bench":
    import bench  # noqa: E402
    sys.argv = ["bench.py"] + sys.argv[3:]
    bench.main()
    sys.exit(0)
from variantcalling_amd.pipelines import filter_variants_pipeline  # noqa: E402

sys.exit(filter_variants_pipeline.run(sys.argv[2:]) or 0)
