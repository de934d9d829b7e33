"""Throughput of the auxiliary hot-path kernels: pileup tally (84 algorithmic B/locus),
SEC likelihood ratio, bridging-SNV test.  Usage: python tools/bench_aux.py"""
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from variantcalling_amd import synth  # noqa: E402
from variantcalling_amd.engine import Engine  # noqa: E402

eng = Engine(0)
out = {}
n_loci = 5_000_000
off, obs = synth.make_pileup(n_loci, seed=5)
eng.upload_pileup(off, obs)
eng.timed_pileup(3)
ms = min(eng.timed_pileup(20) / 20 for _ in range(3))
real = off.size * 8 + obs.size * 2 + n_loci * 40
out["pileup"] = dict(loci=n_loci, observations=int(obs.size), us=round(ms * 1e3, 1), loci_per_s=round(n_loci / (ms * 1e-3)),
                     algorithmic_GBps=round(84.0 * n_loci / (ms * 1e-3) / 1e9, 1), frac_of_8TBps=round(84.0 * n_loci / (ms * 1e-3) / 8e12, 3),
                     actual_bytes_GBps=round(real / (ms * 1e-3) / 1e9, 1))
rng = np.random.default_rng(0)
A = rng.integers(0, 60, size=(2_000_000, 5)).astype(np.int32)
E = rng.integers(0, 400, size=(2_000_000, 5)).astype(np.int32)
t0 = time.perf_counter()
eng.sec_likelihood_ratio(A, E)
out["sec_lr_incl_transfers"] = dict(loci=A.shape[0], wall_ms=round((time.perf_counter() - t0) * 1e3, 1))
# SEC database: build from 8 M cohort observations over 2 M loci, apply to the 5 M resident calls (device only)
from variantcalling_amd import model_io  # noqa: E402
from variantcalling_amd.engine import configure  # noqa: E402
cs = synth.make_callset(5_000_000)
forests = model_io.load_models(os.path.join(ROOT, "tests", "golden", "synth_rf_v1.npz"))["rf_model_ignore_gt_incl_hpol_runs"]
configure(eng, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
eng.upload_variants(cs.variants)
eng.filter_resident()
vk = cs.variants.keys()
loci = np.unique(vk[rng.random(vk.size) < 0.4])
keys = loci[rng.integers(0, loci.size, 8_000_000)]
counts = rng.integers(0, 60, size=(keys.size, 3)).astype(np.int32)
t0 = time.perf_counter()
db_k, db_e = eng.sec_db_build(keys, counts)
t_build = time.perf_counter() - t0
eng.set_sec_db(db_k, db_e)
eng.sec_apply(mark=True, download=False)
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    eng.sec_apply(mark=True, download=False)
    ts.append(time.perf_counter() - t0)
out["sec_db"] = dict(observations=int(keys.size), loci=int(db_k.size), build_wall_ms_incl_transfers=round(t_build * 1e3, 1),
                     variants=cs.variants.n, apply_us_device_plus_launch=round(min(ts) * 1e6, 1),
                     variants_per_s=round(cs.variants.n / min(ts)))
print(json.dumps(out))
