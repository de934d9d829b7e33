"""Where a K1 tile's time goes: core-clock cycles of wave 0 between the kernel's phase boundaries
(kernel variant bit 6), averaged per tile, at 1..4 workgroups per CU.  Phases: 0 tile start -> plan barrier
(column + window + allele/CSR loads landed), 1 -> staged-slices barrier (staging loads, hmer / motif / gc /
cycle-skip, slot atomic), 2 joins, 3 quantise, 4 append, 5 end-of-tile barrier wait.
Usage: python tools/phase3.py [n_variants]"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from variantcalling_amd import model_io, synth  # noqa: E402
from variantcalling_amd.engine import Engine, configure  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
cs = synth.make_callset(n)
forests = model_io.load_models(os.path.join(ROOT, "tests", "golden", "synth_rf_v1.npz"))[
    "rf_model_ignore_gt_incl_hpol_runs"]
eng = Engine(0)
configure(eng, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
eng.upload_variants(cs.variants)
names = ["loads->plan barrier", "staging+window features", "joins", "quantise", "append", "end barrier"]
for bpc in (4, 2, 1):
    v = 1 | 64 | ((bpc & 3) << 12)
    eng.set_kernel_variant(v)
    eng.timed_filter(2)
    eng.phase_clocks(reset=True)
    ms = eng.timed_filter(5) / 5
    c = eng.phase_clocks(reset=True)
    tiles = max(c[6], 1)
    tot = sum(c[:6])
    print(f"== {bpc} workgroups/CU: K0+K1 {ms * 1e3:.1f} us, {tiles} tiles clocked, {tot / tiles:.0f} cycles per tile")
    for k, nm in enumerate(names):
        print(f"   {nm:26s} {c[k] / tiles:9.0f} cyc  {100.0 * c[k] / tot:5.1f} %")
eng.set_kernel_variant(0)
eng.close()
