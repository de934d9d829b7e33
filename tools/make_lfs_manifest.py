"""tests/golden/lfs_manifest.json: sha256 + byte size of every git-LFS fixture of the reference's hot-path tests
(SURVEY.md appendix C), read from the LFS POINTER files of /root/reference/test/resources (the objects themselves are
not pulled there).  tests/test_lfs_replay.py uses it to tell a real fixture from a pointer, wherever the tree is mounted.
Usage: python tools/make_lfs_manifest.py [/root/reference/test/resources]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/test/resources"
DIRS = ["system/test_filter_variants_pipeline", "system/test_train_models_pipeline", "unit/filtering/test_variant_filtering_utils",
        "unit/filtering/test_training_prep", "unit/filtering/test_multiallelics", "unit/filtering/test_spandel",
        "system/test_evaluate_concordance", "system/test_calibrate_bridging_snvs", "general/chr1_head"]
out = {}
for d in DIRS:
    for dirpath, _, files in os.walk(os.path.join(res, d)):
        for f in sorted(files):
            p = os.path.join(dirpath, f)
            rel = os.path.relpath(p, res)
            with open(p, "rb") as fh:
                head = fh.read(200)
            if head.startswith(b"version https://git-lfs.github.com/spec/v1"):
                kv = dict(line.split(" ", 1) for line in head.decode().strip().splitlines())
                out[rel] = dict(sha256=kv["oid"].split(":", 1)[1], size=int(kv["size"]), lfs=True)
            elif os.path.isfile(p) and os.path.getsize(p) < (8 << 20):
                import hashlib
                out[rel] = dict(sha256=hashlib.sha256(open(p, "rb").read()).hexdigest(), size=os.path.getsize(p), lfs=False)
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "lfs_manifest.json"), "w"), indent=1, sort_keys=True)
print(len(out), "fixtures,", sum(v["lfs"] for v in out.values()), "of them LFS objects")
