"""The host-buffer boundary (ugvc_filter_variants on 5 M variants from pageable numpy arrays) timed over chunk counts, in ONE
process so that the counts alternate on the same box and in the same NUMA placement: the boxes of this pool differ by more
between runs (5.1-7.4 ms for the same build) than the settings do.
    python tools/e2e_ab.py [--lib other.so] [--configs 'CHUNKS=8 CHUNKS=16,TAPER=0'] [--rounds 8]
The synthetic callset is cached under /tmp (pickle) so that a second process (another build) starts in seconds."""
import argparse
import os
import pickle
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--configs", default="CHUNKS=16", help="space-separated settings to alternate, each a comma list of "
                    "NAME=value pairs set as UGVC_PIPE_<NAME> before the call, e.g. 'CHUNKS=16,TAPER=0 CHUNKS=16,TAPER=1'")
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--fresh-out", action="store_true", help="allocate the result arrays inside the timed region (np.zeros: "
                    "30 MB of first-touch page faults and an munmap per call) instead of reusing the caller's")
    ap.add_argument("--variants", type=int, default=5_000_000)
    ap.add_argument("--tag", default="new")
    a = ap.parse_args()
    from variantcalling_amd import engine, model_io, synth
    if a.lib:
        engine.load_library(os.path.abspath(a.lib))
    cache = f"/tmp/e2e_ab_callset_{a.variants}.pkl"
    if os.path.exists(cache):
        with open(cache, "rb") as f:
            cs = pickle.load(f)
    else:
        cs = synth.make_callset(a.variants)
        with open(cache, "wb") as f:
            pickle.dump(cs, f, protocol=4)
    forests = model_io.load_models(os.path.join(ROOT, "tests", "golden", "synth_rf_v1.npz"))["rf_model_ignore_gt_incl_hpol_runs"]
    eng = engine.Engine(0)
    engine.configure(eng, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests, "TGCA", 10, 10, True)
    ks = a.configs.split()
    ts = {k: [] for k in ks}
    first = None
    from variantcalling_amd import schema as S
    n = cs.variants.n
    keep = S.FilterResult(np.zeros(n, np.float32), np.zeros(n, np.uint8), np.zeros(n, np.uint8))
    for r in range(a.rounds + 1):
        for k in ks:
            for kv in k.split(","):
                name, val = kv.split("=")
                os.environ["UGVC_PIPE_" + name] = val
            t0 = time.perf_counter()
            res = eng.filter_variants(cs.variants) if a.fresh_out else eng.filter_variants(cs.variants, out=keep)
            dt = (time.perf_counter() - t0) * 1e3
            if first is None:
                first = S.FilterResult(res.tree_score.copy(), res.filter.copy(), res.flags.copy())
            else:
                assert np.array_equal(res.filter, first.filter) and np.array_equal(res.tree_score, first.tree_score)
            if r:
                ts[k].append(dt)
    for k in ks:
        v = np.sort(ts[k])
        print(f"{a.tag} {k}: min {v[0]:.3f} median {np.median(v):.3f} max {v[-1]:.3f} ms  (n={v.size})", flush=True)


if __name__ == "__main__":
    main()
