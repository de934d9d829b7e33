"""How much of the tool's "upload variants + scoring pass + download" stage is first-launch cost (code-object load, lazy allocation)?
One process: context, configure, a 20 k-variant boundary call, then the 5 M-variant call twice."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from variantcalling_amd import model_io, synth
from variantcalling_amd.engine import Engine, configure
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
cs = synth.make_callset(5_000_000)
forests = model_io.load_models(os.path.join(ROOT, "tests", "golden", "synth_rf_v1.npz"))["rf_model_ignore_gt_incl_hpol_runs"]
t0 = time.perf_counter(); eng = Engine(0); t1 = time.perf_counter()
configure(eng, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests); t2 = time.perf_counter()
small = cs.variants.slice(0, 20_000)
eng.filter_variants(small); t3 = time.perf_counter()
eng.filter_variants(small); t4 = time.perf_counter()
eng.filter_variants(cs.variants); t5 = time.perf_counter()
eng.filter_variants(cs.variants); t6 = time.perf_counter()
print(f"context {t1-t0:.3f} s, configure {t2-t1:.3f} s, first 20 k call {t3-t2:.4f} s, second 20 k call {t4-t3:.4f} s, "
      f"first 5 M call {t5-t4:.4f} s, second 5 M call {t6-t5:.4f} s")
eng.close()
