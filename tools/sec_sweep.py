"""SEC apply (bench.py --workload sec_apply) over workgroup sizes and waves per CU in one process: UGVC_SEC_BLOCK x
UGVC_SEC_WAVES are read per launch (profiling knobs of csrc/kernels_sec.hip)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from variantcalling_amd import model_io, synth
    from variantcalling_amd.engine import Engine, configure
    cs = synth.make_callset(5_000_000)
    forests = model_io.load_models(os.path.join(ROOT, "tests", "golden", "synth_rf_v1.npz"))["rf_model_ignore_gt_incl_hpol_runs"]
    eng = Engine(0)
    configure(eng, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
    eng.upload_variants(cs.variants)
    eng.filter_resident()
    rng = np.random.default_rng(0)
    vk = cs.variants.keys()
    loci = np.unique(vk[rng.random(vk.size) < 0.4])
    keys = loci[rng.integers(0, loci.size, 8_000_000)]
    counts = rng.integers(0, 60, size=(keys.size, 3)).astype(np.int32)
    db_k, db_e = eng.sec_db_build(keys, counts)
    eng.set_sec_db(db_k, db_e)
    ref = None
    combos = [(b, w) for b in (512, 1024, 256) for w in (16, 24, 32, 48)]
    for rnd in range(2):
        for b, w in combos:
            os.environ["UGVC_SEC_BLOCK"] = str(b)
            os.environ["UGVC_SEC_WAVES"] = str(w)
            r = eng.sec_apply(mark=False, download=True)
            if ref is None:
                ref = r
            else:
                assert np.array_equal(ref[1], r[1]), (b, w)
            for _ in range(5):
                eng.sec_apply(mark=True, download=False)
            eng.device_sync()
            t0 = time.perf_counter()
            for _ in range(40):
                eng.sec_apply(mark=True, download=False)
            eng.device_sync()
            ms = (time.perf_counter() - t0) / 40 * 1e3
            print(f"block {b} waves/CU {w}: {ms:.4f} ms  frac {26.0 * cs.variants.n / (ms * 1e-3) / 8e12:.3f}", flush=True)


if __name__ == "__main__":
    main()
