"""FASTA -> `schema.Reference` (1 byte/base codes N,A,C,G,T = 0..4), and back.

`--reference_file` of filter_variants_pipeline / `--reference` of train_models_pipeline
(docs/filter_variants_pipeline.md:38-39, docs/train_models_pipeline.md:60).  The reference opens the
FASTA with pyfaidx and fetches per row (pattern: ugvc/pipelines/vcfbed/calibrate_bridging_snvs.py:28-30);
here the whole sequence is encoded once and uploaded to HBM.  Plain or gzip/bgzip files; host logic only."""
from __future__ import annotations

import gzip

import numpy as np

from ..schema import _ASCII_TO_CODE, CODE_TO_CHAR, Reference


def _open(path: str, mode: str = "rb"):
    with open(path, "rb") as fh:
        magic = fh.read(2)
    return gzip.open(path, mode) if magic == b"\x1f\x8b" else open(path, mode)


def read_fasta(path: str, contigs: list | None = None) -> Reference:
    """All records (or only `contigs`, in file order) as one concatenated code array."""
    with _open(path) as fh:
        raw = np.frombuffer(fh.read(), dtype=np.uint8)
    if raw.size == 0 or raw[0] != ord(">"):
        raise ValueError(f"{path}: not a FASTA file")
    nl = np.flatnonzero(raw == 10)
    line_start = np.concatenate([[0], nl + 1])
    line_start = line_start[line_start < raw.size]
    hdr = line_start[raw[line_start] == ord(">")]
    names, parts = [], []
    for k, h in enumerate(hdr):
        e = nl[np.searchsorted(nl, h)] if np.searchsorted(nl, h) < nl.size else raw.size
        name = raw[h + 1:e].tobytes().decode().split()[0] if e > h + 1 else ""
        seq_lo = e + 1
        seq_hi = int(hdr[k + 1]) if k + 1 < hdr.size else raw.size
        if contigs is not None and name not in contigs:
            continue
        seg = raw[seq_lo:seq_hi]
        seg = seg[(seg != 10) & (seg != 13)]
        names.append(name)
        parts.append(_ASCII_TO_CODE[seg])
    if not names:
        raise ValueError(f"{path}: no sequences read")
    off = np.concatenate([[0], np.cumsum([p.size for p in parts])]).astype(np.int64)
    return Reference(np.concatenate(parts) if len(parts) > 1 else parts[0], off, names)


def write_fasta(path: str, ref: Reference, width: int = 60) -> None:
    table = np.frombuffer(CODE_TO_CHAR.encode(), dtype=np.uint8)
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "wb") as fh:
        for c, name in enumerate(ref.names):
            fh.write(f">{name}\n".encode())
            seq = table[ref.codes[ref.contig_off[c]: ref.contig_off[c + 1]]]
            n = seq.size
            full = (n // width) * width
            if full:
                block = np.empty((n // width, width + 1), dtype=np.uint8)
                block[:, :width] = seq[:full].reshape(-1, width)
                block[:, width] = 10
                fh.write(block.tobytes())
            if n > full:
                fh.write(seq[full:].tobytes() + b"\n")


def write_fai(path: str, ref: Reference, width: int = 60) -> None:
    """The samtools `.fai` index beside a FASTA written by write_fasta(path, ref, width) (plain text only): name, length,
    byte offset of the first base, bases per line, bytes per line.  The tools take "--reference_file: Indexed reference
    FASTA file" (docs/filter_variants_pipeline.md:38-39): with the index beside it the contig names are known at once and
    every side-table reader starts together with the FASTA reader instead of after it."""
    if path.endswith(".gz"):
        raise ValueError("write_fai: a gzip FASTA is indexed with a .gzi as well; write a plain FASTA")
    off = 0
    with open(path + ".fai", "w") as fh:
        for c, name in enumerate(ref.names):
            n = int(ref.contig_off[c + 1] - ref.contig_off[c])
            off += len(name) + 2                                   # ">name\n"
            fh.write(f"{name}\t{n}\t{off}\t{width}\t{width + 1}\n")
            off += n + (n + width - 1) // width                   # the bases and one newline per line
