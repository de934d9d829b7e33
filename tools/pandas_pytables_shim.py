"""Run real pandas + PyTables where only an old PyTables build exists (the build container: /opt/conda/bin/python3.9 has
pandas 2.3 and a PyTables 3.6.1 compiled against numpy < 1.20).  PyTables 3.6.1 imports numpy aliases that numpy
1.24+ removed and pandas refuses PyTables < 3.8 by version string; both are patched IN THIS THROWAWAY PROCESS ONLY so
that `DataFrame.to_hdf` / `pd.read_hdf` can serve as the independent implementation the HDF5 layer is checked against
(tools/make_h5_golden.py, tests/test_h5.py).  Never imported by the package.

    /opt/conda/bin/python3.9 tools/pandas_pytables_shim.py write <dir>       # golden inputs written by pandas
    /opt/conda/bin/python3.9 tools/pandas_pytables_shim.py read <file.h5>    # JSON of what pandas reads from a file
"""
import json
import os
import sys
import warnings

warnings.simplefilter("ignore")
import numpy as np  # noqa: E402

np.typeDict = np.sctypeDict
for _n, _t in (("object", object), ("bool", bool), ("int", int), ("float", float), ("str", str), ("complex", complex),
               ("unicode", str), ("long", int)):
    if not hasattr(np, _n):
        setattr(np, _n, _t)
import tables  # noqa: E402

REAL_TABLES_VERSION = tables.__version__
tables.__version__ = "3.8.0"
import pandas as pd  # noqa: E402


def jsonable(v):
    if isinstance(v, (tuple, list)):
        return [jsonable(x) for x in v]
    if v is None or (isinstance(v, float) and v != v):
        return None
    if isinstance(v, (np.bool_, bool)):
        return bool(v)
    if isinstance(v, np.integer):
        return int(v)
    if isinstance(v, np.floating):
        return float(v)
    return v


def frame_json(df):
    if isinstance(df, pd.Series):
        df = df.to_frame(df.name if df.name is not None else "values")
    idx = df.index
    return dict(columns=[str(c) for c in df.columns], dtypes=[str(t) for t in df.dtypes],
                data={str(c): [jsonable(x) for x in df[c].tolist()] for c in df.columns},
                index=[[jsonable(x) for x in idx.get_level_values(i).tolist()] for i in range(idx.nlevels)],
                index_names=[jsonable(x) for x in idx.names])


def write(out):
    rng = np.random.default_rng(7)
    n = 120
    chrom = np.array(["chr%d" % (1 + i * 3 // n) for i in range(n)], dtype=object)
    pos = np.sort(rng.integers(1, 10 ** 6, n)).astype(np.int64)
    d = pd.DataFrame({
        "chrom": chrom, "pos": pos, "ref": rng.choice(list("ACGT"), n),
        "alleles": [("A", "G") if i % 5 else ("AT", "A", "ATT") for i in range(n)],
        "gt_ultima": [(0, 1) if i % 3 else (1, 1) for i in range(n)], "indel": rng.random(n) < 0.3,
        "qual": np.round(rng.random(n) * 90, 2), "sor": rng.random(n).astype(np.float32),
        "dp": rng.integers(0, 80, n).astype(np.int32), "gq": rng.integers(0, 99, n).astype(np.uint8),
        "classify": rng.choice(["tp", "fp", "fn"], n), "tree_score": np.where(rng.random(n) < 0.1, np.nan, rng.random(n)),
        "hmer_indel_length": rng.integers(0, 14, n)})
    d.loc[d["classify"] == "fn", "alleles"] = None
    expected = {}

    def put(path, key, obj, **kw):
        obj.to_hdf(os.path.join(out, path), key=key, **kw)
        expected[f"{path}:{key}"] = frame_json(obj)

    for f in ("pd_fixed.h5", "pd_zlib.h5", "pd_blosc.h5", "pd_table.h5"):
        if os.path.exists(os.path.join(out, f)):
            os.remove(os.path.join(out, f))
    put("pd_fixed.h5", "chr_all", d)                                                   # RangeIndex
    put("pd_fixed.h5", "concordance", d.set_index(["chrom", "pos"], drop=False))       # MultiIndex, as the reference's frames
    put("pd_fixed.h5", "by_ref", d.set_index("ref"))                                   # string index
    put("pd_fixed.h5", "callable_size", pd.Series(np.arange(5.0) * 1.5, name="callable", index=list("abcde")))
    put("pd_fixed.h5", "empty", d.iloc[:0])
    num = d[["pos", "qual", "sor", "dp", "indel"]]
    put("pd_zlib.h5", "num", num, complib="zlib", complevel=5)
    put("pd_blosc.h5", "num", num, complib="blosc", complevel=5)
    tab = d[["chrom", "pos", "qual", "indel", "dp", "classify"]].copy()
    tab.loc[3, "classify"] = np.nan
    put("pd_table.h5", "tab", tab, format="table")
    put("pd_table.h5", "tab_dc", tab, format="table", data_columns=["chrom", "pos"])
    json.dump(dict(pandas=pd.__version__, tables=REAL_TABLES_VERSION, numpy=np.__version__, frames=expected),
              open(os.path.join(out, "pandas_expected.json"), "w"))


def read(path):
    out = {}
    with pd.HDFStore(path, mode="r") as st:
        keys = [k.lstrip("/") for k in st.keys()]
    for k in keys:
        out[k] = frame_json(pd.read_hdf(path, k))
    json.dump(out, sys.stdout)


if __name__ == "__main__":
    {"write": write, "read": read}[sys.argv[1]](sys.argv[2])
