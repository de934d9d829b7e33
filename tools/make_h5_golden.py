"""Golden vectors for variantcalling_amd/io/h5.py, made with libhdf5's own tools (run in the build container, where
/opt/conda/bin/h5dump, h5repack and /root/reference exist; the outputs under tests/golden/h5/ are committed).

1. reference_fixture.json - what libhdf5 (`h5dump`) sees in the one real pandas/PyTables file of the reference tree
   (test/resources/unit/comparison/test_vcf_pipeline_utils/annotate_concordance_h5_input.hdf): object names, dataset
   shapes / type classes, the string datasets' contents, every scalar attribute.
2. frame_contig.h5 (written by io.h5.write_hdf), frame_gzip.h5 / frame_gzip_only.h5 / frame_fletcher.h5 (the same
   objects re-created by libhdf5 through `h5repack`: chunked + shuffle + deflate, deflate only, fletcher32) and
   frame_expected.npz (the columns the files hold).
3. pd_fixed / pd_zlib / pd_blosc / pd_table .h5.gz + pandas_expected.json - files written by REAL pandas + PyTables
   (`DataFrame.to_hdf`, fixed and table format, zlib and blosc compression) through tools/pandas_pytables_shim.py in
   /opt/conda's interpreter, gzip-ed because PyTables pre-allocates 1 MiB per pickled object block.
Usage: python tools/make_h5_golden.py"""
import json
import os
import re
import subprocess
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from variantcalling_amd.io import h5  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "h5")
BIN = "/opt/conda/bin"
FIX = "/root/reference/test/resources/unit/comparison/test_vcf_pipeline_utils/annotate_concordance_h5_input.hdf"


def h5dump(*args):
    r = subprocess.run([os.path.join(BIN, "h5dump"), *args], capture_output=True, text=True)
    if r.returncode:
        raise SystemExit(r.stderr)
    return r.stdout


def reference_fixture():
    ls = subprocess.run([os.path.join(BIN, "h5ls"), "-r", FIX], capture_output=True, text=True).stdout
    objs = {}
    for line in ls.splitlines():
        m = re.match(r"(\S+)\s+(Group|Dataset)\s*(\{.*\})?", line)
        if m:
            objs[m.group(1)] = dict(kind=m.group(2), shape=m.group(3))
    strings = {}
    for d, o in objs.items():
        if o["kind"] != "Dataset":
            continue
        hdr = h5dump("-H", "-d", d, FIX)
        o["type"] = re.search(r"DATATYPE\s+(H5T_\w+)", hdr).group(1)
        if o["type"] == "H5T_STRING":
            txt = h5dump("-y", "-w", "0", "-d", d, FIX)
            strings[d] = re.findall(r'"([^"]*)"', re.search(r"DATA \{(.*?)\n   \}", txt, re.S).group(1))
    attrs = {}
    txt = h5dump("-A", FIX)
    cur = None
    for m in re.finditer(r'(GROUP|DATASET) "([^"]+)"|ATTRIBUTE "([^"]+)" \{\s*DATATYPE\s+([^\n]+)(?:[^}]*?\})?\s*DATASPACE\s+([^\n]+)\s*DATA \{\s*([^}]*)\}', txt):
        if m.group(1):
            cur = m.group(2)
            attrs[cur] = {}
        else:
            v = re.sub(r"^\(0\):\s*", "", " ".join(m.group(6).split()))
            attrs[cur][m.group(3)] = v.strip('"') if v.startswith('"') else v
    json.dump(dict(source=FIX, objects=objs, string_datasets=strings, attributes=attrs),
              open(os.path.join(OUT, "reference_fixture.json"), "w"), indent=1, sort_keys=True)


def frames():
    rng = np.random.default_rng(20260923)
    n = 300
    chrom = np.array([f"chr{1 + i * 3 // n}" for i in range(n)], dtype=object)
    pos = np.sort(rng.integers(1, 5_000_000, n)).astype(np.int64)
    cols = [("chrom", chrom), ("pos", pos), ("ref", np.array(list("ACGT"), dtype=object)[rng.integers(0, 4, n)]),
            ("indel", rng.random(n) < 0.2), ("qual", np.round(rng.random(n) * 100, 2)),
            ("sor", rng.random(n).astype(np.float32)), ("dp", rng.integers(0, 90, n).astype(np.int32)),
            ("classify", np.array(["tp", "fp", "fn"], dtype=object)[rng.integers(0, 3, n)]),
            ("tree_score", np.where(rng.random(n) < 0.05, np.nan, rng.random(n))),
            ("hmer_indel_length", rng.integers(0, 14, n).astype(np.int64)), ("gq", rng.integers(0, 99, n).astype(np.uint8))]
    fr = h5.Frame(cols, index=[chrom, pos], index_names=["chrom", "pos"])
    small = h5.Frame([("group", np.array(["SNP", "INDELS"], dtype=object)), ("tp", np.array([5, 7], np.int64)),
                      ("precision", np.array([0.5, 0.25]))])
    src = os.path.join(OUT, "frame_contig.h5")
    h5.write_hdf(src, {"concordance": fr, "optimal_recall_precision": small,
                       "empty": h5.Frame([("a", np.zeros(0)), ("s", np.zeros(0, object))])})
    for name, flt in (("frame_gzip", ["-f", "SHUF", "-f", "GZIP=6", "-l", "CHUNK=100x1"]), ("frame_gzip_only", ["-f", "GZIP=1"]),
                      ("frame_fletcher", ["-f", "FLET"])):
        dst = os.path.join(OUT, name + ".h5")
        if os.path.exists(dst):
            os.remove(dst)
        subprocess.run([os.path.join(BIN, "h5repack"), *flt, src, dst], check=True)
    np.savez_compressed(os.path.join(OUT, "frame_expected.npz"), **{k: (v.astype("U") if v.dtype == object else v) for k, v in cols})


CONDA_PY = "/opt/conda/bin/python3.9"


def pandas_files():
    import gzip
    import shutil
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run([CONDA_PY, os.path.join(ROOT, "tools", "pandas_pytables_shim.py"), "write", tmp], check=True, cwd=tmp)
        for f in sorted(os.listdir(tmp)):
            if f.endswith(".h5"):
                with open(os.path.join(tmp, f), "rb") as src, gzip.GzipFile(os.path.join(OUT, f + ".gz"), "wb", 9, mtime=0) as dst:
                    shutil.copyfileobj(src, dst)
            else:
                shutil.copy(os.path.join(tmp, f), os.path.join(OUT, f))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    reference_fixture()
    frames()
    pandas_files()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
