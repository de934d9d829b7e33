import os, sys
sys.path.insert(0, os.getcwd())
from variantcalling_amd import model_io, synth
from variantcalling_amd.engine import Engine, configure
forests = model_io.load_models("tests/golden/synth_rf_v1.npz")["rf_model_ignore_gt_incl_hpol_runs"]
eng = Engine(0)
full = synth.make_callset(5_000_000)
configure(eng, full.ref, full.runs, full.tracks, full.blacklist, forests)
eng.upload_variants(full.variants)
for v in (1, 1 | 131072):
    eng.set_kernel_variant(v)
    eng.timed_filter(3)
eng.close()
