"""RTG SDF `seqdata0` decoder (TEST INFRASTRUCTURE - only used to mint golden fixtures).

The reference repo ships a real hg38 slice only as an RTG "SDF" directory
(`test/resources/general/chr1_head/Homo_sapiens_assembly38.fasta.sdf/seqdata0`,
`test/resources/general/sample.fasta.sdf/seqdata0`); the FASTA files themselves are
un-pulled git-LFS pointers.  Layout (verified by MD5 against the
`M5` tags of the `.dict` files next to them): every block of 64 residues is three
consecutive big-endian uint64 words = bit planes b2, b1, b0; residue i of the block
is bit i (LSB first) of each word; code = 4*b2 + 2*b1 + b0 with N=0 A=1 C=2 G=3 T=4,
which is the base alphabet the whole engine uses (variantcalling_amd/schema.py).
"""
from __future__ import annotations

import hashlib

import numpy as np

CODE_TO_ASCII = np.frombuffer(b"NACGT", dtype=np.uint8)


def decode_seqdata(path: str, n_residues: int) -> np.ndarray:
    """Return the u8 code array (N,A,C,G,T = 0..4) stored in an SDF `seqdata0` file."""
    raw = np.fromfile(path, dtype=">u8")
    n_blocks = (n_residues + 63) // 64
    if raw.size < 3 * n_blocks:
        raise ValueError(f"{path}: {raw.size} words, need {3 * n_blocks}")
    planes = raw[: 3 * n_blocks].reshape(n_blocks, 3).astype(np.uint64)
    shifts = np.arange(64, dtype=np.uint64)
    bits = (planes[:, :, None] >> shifts[None, None, :]) & np.uint64(1)  # [blk, plane, bit]
    codes = (bits[:, 0, :] * 4 + bits[:, 1, :] * 2 + bits[:, 2, :]).astype(np.uint8)
    return codes.reshape(-1)[:n_residues].copy()


def md5_of_codes(codes: np.ndarray) -> str:
    return hashlib.md5(CODE_TO_ASCII[codes].tobytes()).hexdigest()
