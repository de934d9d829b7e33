"""Worker of tests/test_vcf_native.py::test_corrupted_inputs_never_crash: feeds byte-mutated VCF / BGZF / FASTA / BED /
HDF5 files to the native codec and the HDF5 reader in THIS process; the parent only looks at the exit status (a
segfault or abort in libugvc_vcf.so would kill the process) and at the tally printed at the end."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from variantcalling_amd.io import h5, vcf_native as nv  # noqa: E402

src_dir, seed, rounds = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(seed)
names = ["chr1", "chr2", "chr3"]
files = sorted(f for f in os.listdir(src_dir) if not f.startswith("mut"))
ok = err = 0
for r in range(rounds):
    f = files[r % len(files)]
    data = bytearray(open(os.path.join(src_dir, f), "rb").read())
    kind = int(rng.integers(0, 4))
    if kind == 0 and len(data) > 8:                                   # flip a few bytes
        for _ in range(int(rng.integers(1, 6))):
            data[int(rng.integers(0, len(data)))] = int(rng.integers(0, 256))
    elif kind == 1 and len(data) > 8:                                 # truncate
        del data[int(rng.integers(0, len(data))):]
    elif kind == 2 and len(data) > 64:                                # duplicate / drop a span
        a = int(rng.integers(0, len(data) - 32))
        b = a + int(rng.integers(1, 32))
        if rng.random() < 0.5:
            data[a:a] = data[a:b]
        else:
            del data[a:b]
    else:                                                             # splice in noise
        a = int(rng.integers(0, len(data) + 1))
        data[a:a] = bytes(rng.integers(0, 256, int(rng.integers(1, 64)), dtype=np.uint8))
    path = os.path.join(src_dir, "mut_" + f)
    with open(path, "wb") as fh:
        fh.write(data)
    try:
        if f.endswith((".vcf", ".vcf.gz")):
            v = nv.read_vcf(path, names, is_mutect=bool(r & 1))
            if v.table.n and r % 3 == 0:                              # and push a parsed file through the writer
                res = type("R", (), {})()
                from variantcalling_amd import schema as S
                res = S.FilterResult(np.zeros(v.table.n, np.float32), np.zeros(v.table.n, np.uint8), np.zeros(v.table.n, np.uint8))
                nv.write_filtered_vcf(os.path.join(src_dir, "mut_out.vcf.gz"), v, res)
        elif f.endswith((".fa", ".fa.gz")):
            nv.read_fasta(path)
        elif f.endswith((".bed", ".interval_list")):
            nv.read_intervals(path, names, merge=bool(r & 1))
        elif f.endswith(".h5"):
            with h5.H5File(path) as hf:
                for k in hf.keys():
                    try:
                        h5.read_hdf(path, k)
                    except (h5.H5Error, KeyError, ValueError, IndexError, OverflowError, MemoryError, EOFError,
                            UnicodeDecodeError, TypeError, AttributeError, ImportError, Exception):
                        err += 1
        ok += 1
    except Exception:
        err += 1
print(f"done ok={ok} err={err}")
