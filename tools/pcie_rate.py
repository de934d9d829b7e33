"""PCIe-inclusive rate of the host-buffer boundary (`ugvc_filter_variants`: upload the variant columns, one scoring
pass, download score / FILTER / flags) next to the resident rate bench.py reports.  Usage: python tools/pcie_rate.py [n]"""
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from variantcalling_amd import model_io, synth  # noqa: E402
from variantcalling_amd.engine import Engine, configure  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
forests = model_io.load_models(os.path.join(ROOT, "tests", "golden", "synth_rf_v1.npz"))["rf_model_ignore_gt_incl_hpol_runs"]
cs = synth.make_callset(n)
vt = cs.variants
eng = Engine(0)
configure(eng, cs.ref, cs.runs, cs.tracks, cs.blacklist, forests)
eng.filter_variants(vt)
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    eng.filter_variants(vt)
    ts.append(time.perf_counter() - t0)
up = sum(getattr(vt, c).nbytes for c in vt.COLS if c != "gt") + vt.alleles.nbytes
down = vt.n * 6
resident = eng.timed_filter(20) / 20
print(json.dumps(dict(variants=vt.n, host_to_device_MB=round(up / 1e6, 1), device_to_host_MB=round(down / 1e6, 1),
                      wall_ms=round(min(ts) * 1e3, 2), variants_per_s_pcie_inclusive=round(vt.n / min(ts)),
                      resident_pass_ms=round(resident, 3), variants_per_s_resident=round(vt.n / (resident * 1e-3)))))
eng.close()
