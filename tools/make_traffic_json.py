"""Turn rocprofv3 PMC passes (tools/gpu_evidence.sh prof) into one entry of profiles/hbm_traffic.json.

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE summed over the kernels of the workload (counter unit: KB).  The
factor 2 is the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE tallies 128-byte requests at 64
bytes), re-measured here on known byte counts (tools/calib, profiles/r01_calibration.txt: 0.500 x for 4- and
16-byte/lane streaming reads; WRITE_SIZE 1.000 x).

The file is keyed by what was measured - "<workload>:<units>:<world>", e.g. "filter:4999706:1", "c2:1000000:1",
"pileup:5000000:1", "sec_apply:4999706:1", "c5_feature_build:2000000:1" - and bench.py prints `roofline.traffic` only for the
key it is running (the 5 M-pass figure used to appear on every line).

Usage: python tools/make_traffic_json.py <prof dir> --key filter:4999706:1 [--kernels substr,substr] [--dst profiles/hbm_traffic.json]"""
import argparse
import csv
import glob
import json
import os

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--key", required=True)
ap.add_argument("--kernels", default="", help="comma-separated substrings: only kernels whose name contains one of them count")
ap.add_argument("--dst", default=os.path.join(os.path.dirname(__file__), "..", "profiles", "hbm_traffic.json"))
a = ap.parse_args()
want = [k for k in a.kernels.split(",") if k]
per = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(a.dir, f"pmc_{c}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")       # template instantiations print a return type
            if k.startswith("ugvc::") and (not want or any(w in k for w in want)):
                per.setdefault(k, {}).setdefault(c, []).append(float(r["Counter_Value"]))
head = os.path.join(os.path.dirname(__file__), "..", "profiles", "HEAD")
entry = {"source": a.dir, "commit": open(head).read().strip() if os.path.exists(head) else None,
         "unit": "HBM bytes per launch of the workload's kernels: 2 x FETCH_SIZE + WRITE_SIZE (KB counters), mean over the traced launches",
         "kernels": {}}
tot = 0.0
for k, cs in sorted(per.items()):
    fk = sum(cs.get("FETCH_SIZE", [0])) / max(1, len(cs.get("FETCH_SIZE", [0])))
    wk = sum(cs.get("WRITE_SIZE", [0])) / max(1, len(cs.get("WRITE_SIZE", [0])))
    b = 2.0 * fk * 1024 + wk * 1024
    entry["kernels"][k] = {"FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "hbm_bytes": b, "launches_traced": len(cs.get("FETCH_SIZE", []))}
    tot += b
entry["bytes_per_launch"] = tot if per else None
doc = {}
if os.path.exists(a.dst):
    try:
        doc = json.load(open(a.dst))
    except Exception:
        doc = {}
if "workloads" not in doc:
    doc = {"fetch_correction": 2.0, "workloads": {}}
doc["workloads"][a.key] = entry
json.dump(doc, open(a.dst, "w"), indent=1)
print(a.key, json.dumps(entry, indent=1))
