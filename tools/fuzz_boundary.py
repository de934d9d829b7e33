"""Fuzz of the host-buffer boundary call (csrc/pipeline.hip) against the plain sequence upload -> resident pass -> download on
the same context: random callset sizes above the pipeline threshold, chunk counts, tapering, canonical / swapped / gapped
allele pools, fresh / reused result arrays; afterwards the resident state (a second resident pass, the feature matrix) must
agree too.  python tools/fuzz_boundary.py [rounds] [seed]"""
import copy
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def relayout(vt, rng, mode):
    """The same alleles in a pool laid out differently: 'swap' = ALT before REF, 'gap' = canonical with random gaps."""
    rl, al = vt.ref_len.astype(np.int64), vt.alt_len.astype(np.int64)
    out = copy.copy(vt)
    if mode == "swap":
        start = np.cumsum(rl + al) - (rl + al)
        a_off, r_off, total = start, start + al, int((rl + al).sum())
    else:
        gaps = (rng.random(vt.n) < 0.001).astype(np.int64) * rng.integers(1, 5, vt.n)
        start = np.cumsum(rl + al + gaps) - (rl + al)
        r_off, a_off, total = start, start + rl, int((rl + al + gaps).sum())
    pool = np.zeros(total, np.uint8)
    for k in range(int(max(rl.max(), al.max()))):
        m = np.flatnonzero(al > k)
        pool[a_off[m] + k] = vt.alleles[vt.alt_off[m].astype(np.int64) + k]
        m = np.flatnonzero(rl > k)
        pool[r_off[m] + k] = vt.alleles[vt.ref_off[m].astype(np.int64) + k]
    out.alleles, out.ref_off, out.alt_off = pool, r_off.astype(np.uint32), a_off.astype(np.uint32)
    return out


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    from variantcalling_amd import model_io, schema as S, synth
    from variantcalling_amd.engine import Engine, configure
    rng = np.random.default_rng(seed)
    forests = model_io.load_models(os.path.join(ROOT, "tests", "golden", "synth_rf_v1.npz"))["rf_model_ignore_gt_incl_hpol_runs"]
    eng = Engine(0)
    bad = 0
    for r in range(rounds):
        n = int(rng.integers(262_144, 700_000))
        cs = synth.make_callset(n, genome_len=int(rng.integers(40_000_000, 400_000_000)), n_contigs=int(rng.integers(1, 9)), seed=int(rng.integers(1, 10**6)))
        configure(eng, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests, "TGCA", 10, 10, True)
        vt = cs.variants
        os.environ["UGVC_PIPE_CHUNKS"] = "1"
        ref = eng.filter_variants(vt)                                   # plain upload / pass / download
        X_ref, g_ref = eng.feature_matrix()
        for mode in ("canonical", "swap", "gap"):
            t = vt if mode == "canonical" else relayout(vt, rng, mode)
            os.environ["UGVC_PIPE_CHUNKS"] = str(int(rng.integers(2, 24)))
            os.environ["UGVC_PIPE_TAPER"] = str(int(rng.integers(0, 2)))
            os.environ["UGVC_PIPE_DERIVE"] = str(int(rng.random() < 0.8))
            keep = S.FilterResult(np.full(t.n, 3, np.float32), np.full(t.n, 3, np.uint8), np.full(t.n, 3, np.uint8))
            got = eng.filter_variants(t, out=keep) if rng.random() < 0.5 else eng.filter_variants(t)
            ok = np.array_equal(got.filter, ref.filter) and np.array_equal(got.flags, ref.flags) and np.array_equal(got.tree_score, ref.tree_score)
            eng.filter_resident()
            again = eng.download_results()
            ok2 = np.array_equal(again.filter, ref.filter) and np.array_equal(again.tree_score, ref.tree_score)
            X, g = eng.feature_matrix()
            ok3 = np.array_equal(X, X_ref) and np.array_equal(g, g_ref)
            if not (ok and ok2 and ok3):
                bad += 1
            print(f"round {r} n {t.n} {mode:9s} chunks {os.environ['UGVC_PIPE_CHUNKS']:>2s} taper {os.environ['UGVC_PIPE_TAPER']} derive {os.environ['UGVC_PIPE_DERIVE']}: "
                  f"call {'ok' if ok else 'DIFFERS'}, resident pass {'ok' if ok2 else 'DIFFERS'}, feature matrix {'ok' if ok3 else 'DIFFERS'}", flush=True)
    print("mismatching cases:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
