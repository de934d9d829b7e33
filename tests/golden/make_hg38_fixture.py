"""Mint tests/golden/hg38_chr1_head.npz from the reference repo's RTG SDF fixture.

Runs only in the build container (needs /root/reference).  The output is the REAL
hg38 chr1:1-5,000,000 sequence (MD5 f00eb808cff4be46d8c69c7209038873, the `M5` tag of
test/resources/general/chr1_head/Homo_sapiens_assembly38.dict:2) and the chr20 100 kb
sample (MD5 042e5a811f0a907d7e9f63e558c70f75, test/resources/general/sample.fasta.dict:2),
stored as 2 bits/base (A,C,G,T = 0..3) plus the list of N runs, so it can travel to the
GPU box where /root/reference does not exist.  Usage:  python tests/golden/make_hg38_fixture.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle.sdf import decode_seqdata, md5_of_codes  # noqa: E402

REF = "/root/reference/test/resources/general"
SRC = {
    "chr1": (f"{REF}/chr1_head/Homo_sapiens_assembly38.fasta.sdf/seqdata0", 5_000_000,
             "f00eb808cff4be46d8c69c7209038873"),
    "chr20": (f"{REF}/sample.fasta.sdf/seqdata0", 100_000, "042e5a811f0a907d7e9f63e558c70f75"),
}


def pack(codes):
    isn = (codes == 0).astype(np.int8)
    d = np.diff(np.concatenate([[0], isn, [0]]))
    runs = np.stack([np.where(d == 1)[0], np.where(d == -1)[0]], axis=1).astype(np.int64)
    two = np.where(codes == 0, 0, codes - 1).astype(np.uint8)
    pad = (-two.size) % 4
    two = np.concatenate([two, np.zeros(pad, np.uint8)]).reshape(-1, 4)
    packed = (two[:, 0] | (two[:, 1] << 2) | (two[:, 2] << 4) | (two[:, 3] << 6)).astype(np.uint8)
    return packed, runs


out = {}
for name, (path, n, md5) in SRC.items():
    codes = decode_seqdata(path, n)
    assert md5_of_codes(codes) == md5, name
    packed, runs = pack(codes)
    out[f"{name}_packed"] = packed
    out[f"{name}_nruns"] = runs
    out[f"{name}_len"] = np.int64(n)
    out[f"{name}_md5"] = np.frombuffer(md5.encode(), dtype=np.uint8)
dst = os.path.join(os.path.dirname(__file__), "hg38_chr1_head.npz")
np.savez_compressed(dst, **out)
print(dst, os.path.getsize(dst))
