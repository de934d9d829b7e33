"""Turn the rocprofv3 PMC passes of tools/gpu_profile.sh into profiles/hbm_traffic.json.

HBM bytes per scoring pass = 2 x FETCH_SIZE + WRITE_SIZE summed over the kernels of the pass (counter
unit: KB).  The factor 2 is the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE tallies
128-byte requests at 64 bytes), re-measured here on known byte counts (tools/calib, profiles/r01_calibration.txt:
0.500 x for 4- and 16-byte/lane streaming reads; WRITE_SIZE 1.000 x).
Usage: python tools/make_traffic_json.py gpurun_out/prof_<tag> [profiles/hbm_traffic.json]"""
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(__file__), "..", "profiles", "hbm_traffic.json")
per = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(d, f"pmc_{c}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")       # template instantiations print a return type
            if k.startswith("ugvc::"):
                per.setdefault(k, {}).setdefault(c, []).append(float(r["Counter_Value"]))
head = os.path.join(os.path.dirname(__file__), "..", "profiles", "HEAD")
out = {"source": d, "commit": open(head).read().strip() if os.path.exists(head) else None,
       "unit": "bytes per scoring pass (5 M variants, 1 GPU)", "fetch_correction": 2.0, "kernels": {}}
tot = 0.0
for k, cs in sorted(per.items()):
    fk = sum(cs.get("FETCH_SIZE", [0])) / max(1, len(cs.get("FETCH_SIZE", [0])))
    wk = sum(cs.get("WRITE_SIZE", [0])) / max(1, len(cs.get("WRITE_SIZE", [0])))
    b = 2.0 * fk * 1024 + wk * 1024
    out["kernels"][k] = {"FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "hbm_bytes": b}
    tot += b
out["bytes_per_launch_5M"] = tot
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
