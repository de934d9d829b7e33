"""HDF5 without libhdf5: the subset PyTables / pandas write.

The reference moves labelled call tables between its tools as pandas HDF5 files: `train_models_pipeline
--input_file x.h5` is "output of comparison" (/root/reference/docs/train_models_pipeline.md:44-46),
`evaluate_concordance` reads `pd.read_hdf(input_file, key)` and writes `optimal_recall_precision` /
`recall_precision_curve` keys (/root/reference/ugvc/pipelines/evaluate_concordance.py:57-69,99-108), the reports load
`pd.read_hdf(h5, key=...)` (/root/reference/ugvc/reports/report_data_loader.py:37-60).  pandas delegates to PyTables,
which is not installable here, so this module speaks the file format itself (HDF5 File Format Specification 1.x /
2.0: superblock 0-3, version-1 object headers, symbol-table groups, B-tree v1 chunk indices, local and global heaps,
deflate / shuffle / fletcher32 filters) and, on top of it, pandas' "fixed" layout (`pandas_type` = frame / series:
axis0, axis1, block<i>_items, block<i>_values; object blocks are one pickled ndarray in a variable-length row, empty
arrays are a dummy element plus `shape` / `value_type` attributes).

    cols = read_hdf("calls.h5", key="concordance")     # Frame: ordered column name -> numpy array, plus .index
    write_hdf("out.h5", {"optimal_recall_precision": Frame(...)})

Pinned against the only real (non-LFS) HDF5 file in the reference tree
(test/resources/unit/comparison/test_vcf_pipeline_utils/annotate_concordance_h5_input.hdf, a 25-column concordance
frame written by pandas 0.15.2-format / PyTables 2.1) and, for files produced by `write_hdf`, against libhdf5's own
`h5dump` where that tool exists (tests/test_h5.py).  Host-side I/O only: nothing here touches the GPU path.
"""
from __future__ import annotations

import mmap
import pickle
import struct
import zlib
from collections import OrderedDict

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(Exception):
    pass


# ------------------------------------------------------------------------------------------------ datatypes
class _Type:
    """Decoded datatype message: `dtype` is the numpy dtype of one stored element (16-byte descriptors for
    variable-length data), `kind` in {"num", "bool", "str", "vlen", "vlen_str", "compound", "enum", "opaque"}."""

    def __init__(self, kind, dtype, size, base=None, cset=0, enum=None):
        self.kind, self.dtype, self.size, self.base, self.cset, self.enum = kind, dtype, size, base, cset, enum


def _parse_type(buf, pos, osz=8):
    cv, b0, b1, b2, size = struct.unpack_from("<BBBBI", buf, pos)
    cls, ver = cv & 15, cv >> 4
    p = pos + 8
    if cls == 0:                                                   # fixed point
        order = ">" if b0 & 1 else "<"
        signed = bool(b0 & 8)
        return _Type("num", np.dtype(f"{order}{'i' if signed else 'u'}{size}"), size), p + 4
    if cls == 1:                                                   # float
        order = ">" if b0 & 1 else "<"
        return _Type("num", np.dtype(f"{order}f{size}"), size), p + 12
    if cls == 2:                                                   # time
        return _Type("num", np.dtype(f"<i{size}"), size), p + 2
    if cls == 3:                                                   # fixed-length string
        return _Type("str", np.dtype(f"S{size}"), size, cset=b0 >> 4), p
    if cls == 4:                                                   # bitfield: PyTables' bool is B8
        return _Type("bool" if size == 1 else "num", np.dtype(np.bool_ if size == 1 else f"<u{size}"), size), p + 4
    if cls == 5:                                                   # opaque
        taglen = b0
        return _Type("opaque", np.dtype(f"V{size}"), size), p + ((taglen + 7) & ~7)
    if cls == 6:                                                   # compound
        n = b0 | (b1 << 8)
        names, formats, offsets = [], [], []
        for _ in range(n):
            e = buf.find(b"\0", p)
            name = bytes(buf[p:e]).decode()
            if ver < 3:
                p += (e - p + 8) & ~7                              # name + NUL padded to 8
                off = struct.unpack_from("<I", buf, p)[0]
                p += 4
                if ver == 1:
                    ndim = buf[p]
                    dims = struct.unpack_from("<4I", buf, p + 12)[:ndim]
                    p += 28
            else:
                p = e + 1
                nb = max(1, (size.bit_length() + 7) // 8)
                off = int.from_bytes(bytes(buf[p:p + nb]), "little")
                p += nb
                dims = ()
            mt, p = _parse_type(buf, p, osz)
            if mt.kind not in ("num", "bool", "str", "enum", "opaque", "compound"):
                raise H5Error(f"compound member {name!r} of kind {mt.kind} is not supported")
            names.append(name)
            formats.append((mt.dtype, tuple(int(d) for d in dims)) if ver == 1 and dims else mt.dtype)
            offsets.append(off)
        return _Type("compound", np.dtype(dict(names=names, formats=formats, offsets=offsets, itemsize=size)), size), p
    if cls == 7:                                                   # reference
        return _Type("num", np.dtype(f"<u{size}"), size), p
    if cls == 8:                                                   # enum
        n = b0 | (b1 << 8)
        base, p = _parse_type(buf, p, osz)
        names = []
        for _ in range(n):
            e = buf.find(b"\0", p)
            names.append(bytes(buf[p:e]).decode())
            p = p + ((e - p + 8) & ~7) if ver < 3 else e + 1
        vals = np.frombuffer(bytes(buf[p:p + n * base.size]), base.dtype, n)
        p += n * base.size
        mapping = dict(zip(names, (int(v) for v in vals)))
        if set(mapping) == {"FALSE", "TRUE"} and base.size == 1:   # h5py's bool
            return _Type("bool", np.dtype(np.bool_), 1, enum=mapping), p
        return _Type("enum", base.dtype, base.size, enum=mapping), p
    if cls == 9:                                                   # variable length
        base, p = _parse_type(buf, p, osz)
        is_str = (b0 & 15) == 1
        return _Type("vlen_str" if is_str else "vlen", np.dtype(f"V{size}"), size, base=base, cset=b1 & 15), p
    if cls == 10:                                                  # array
        ndim = buf[p]
        p += 4 if ver == 2 else 1
        dims = struct.unpack_from(f"<{ndim}I", buf, p)
        p += 4 * ndim
        if ver == 2:
            p += 4 * ndim
        base, p = _parse_type(buf, p, osz)
        return _Type(base.kind, np.dtype((base.dtype, tuple(int(d) for d in dims))), size, enum=base.enum), p
    raise H5Error(f"datatype class {cls} is not supported")


# ------------------------------------------------------------------------------------------------ reader
class _Node:
    """An object header: messages decoded lazily into .attrs / dataset metadata."""

    def __init__(self, f, addr, name):
        self.file, self.addr, self.name = f, addr, name
        self._msgs = f._read_header(addr)
        self._attrs = None

    @property
    def attrs(self):
        if self._attrs is None:
            self._attrs = OrderedDict()
            for t, _, pos, size in self._msgs:
                if t == 0x000C:
                    k, v = self.file._read_attribute(pos, size)
                    self._attrs[k] = v
        return self._attrs

    def _msg(self, t):
        for m in self._msgs:
            if m[0] == t:
                return m
        return None

    @property
    def is_group(self):
        return self._msg(0x0011) is not None


class Group(_Node):
    def __init__(self, f, addr, name):
        super().__init__(f, addr, name)
        self._links = None

    def _load(self):
        if self._links is None:
            m = self._msg(0x0011)
            if m is None:
                raise H5Error(f"{self.name}: groups stored as link messages (new-style) are not supported")
            bt, heap = self.file._unpack_addrs(m[2], 2)
            self._links = OrderedDict(sorted(self.file._walk_group_btree(bt, heap)))
        return self._links

    def keys(self):
        return list(self._load().keys())

    def __contains__(self, k):
        return k in self._load()

    def __getitem__(self, path):
        node = self
        for part in [p for p in path.split("/") if p]:
            links = node._load()
            if part not in links:
                raise KeyError(f"{path!r}: no object {part!r} in {node.name!r}")
            child = (node.name.rstrip("/") + "/" + part)
            probe = _Node(self.file, links[part], child)
            node = Group(self.file, links[part], child) if probe.is_group else Dataset(self.file, links[part], child)
        return node


class Dataset(_Node):
    def __init__(self, f, addr, name):
        super().__init__(f, addr, name)
        m = self._msg(0x0001)
        self.shape, self.maxshape = f._parse_dataspace(m[2]) if m else ((), ())
        m = self._msg(0x0003)
        if m is None:
            raise H5Error(f"{name}: no datatype message")
        if m[1] & 2:
            raise H5Error(f"{name}: committed (shared) datatypes are not supported")
        self.type, _ = _parse_type(f.buf, m[2], f.osz)
        self.filters = f._parse_filters(self._msg(0x000B))

    @property
    def dtype(self):
        return self.type.dtype

    def read(self):
        """The whole dataset: numpy array of `shape`; variable-length rows come back as an object array of
        uint8 arrays (or str)."""
        f = self.file
        m = self._msg(0x0008)
        if m is None:
            raise H5Error(f"{self.name}: no layout message")
        n = int(np.prod(self.shape, dtype=np.int64)) if self.shape is not None else 0
        esz = self.type.size
        raw = f._read_layout(m[2], self.shape, esz, self.filters, self.name)
        if raw is None:                                            # never allocated: fill value (zeros)
            raw = bytes(n * esz)
        return f._decode(self.type, self.shape, raw[:n * esz])


class H5File:
    """Read-only view of an HDF5 file.  `H5File(path)["/group/dataset"].read()`."""

    def __init__(self, path):
        self.path = path
        self._fh = open(path, "rb")
        try:
            self.buf = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
        except ValueError:
            self._fh.close()
            raise H5Error(f"{path}: empty file")
        self._gcol = {}
        base = 0
        while True:
            if self.buf[base:base + 8] == SIGNATURE:
                break
            base = 512 if base == 0 else base * 2
            if base + 8 > len(self.buf):
                self.close()
                raise H5Error(f"{path}: not an HDF5 file (no superblock signature)")
        b = self.buf
        ver = b[base + 8]
        if ver in (0, 1):
            self.osz, self.lsz = b[base + 13], b[base + 14]
            self.leaf_k, self.int_k = struct.unpack_from("<HH", b, base + 16)
            p = base + 24 + (4 if ver == 1 else 0)
            self.base = self._addr(p)
            p += 4 * self.osz                                      # base, free-space, EOF, driver block
            root_hdr = self._addr(p + self.osz)                    # symbol table entry: name offset, header address
        elif ver in (2, 3):
            self.osz, self.lsz = b[base + 9], b[base + 10]
            self.leaf_k, self.int_k = 4, 16
            self.base = self._addr(base + 12)
            root_hdr = self._addr(base + 12 + 3 * self.osz)
        else:
            self.close()
            raise H5Error(f"{path}: superblock version {ver} is not supported")
        if self.osz not in (4, 8) or self.lsz not in (4, 8):
            self.close()
            raise H5Error(f"{path}: offset/length sizes {self.osz}/{self.lsz} are not supported")
        if self.base == UNDEF:
            self.base = 0
        self.root = Group(self, root_hdr + self.base, "/")

    # -- context manager
    def close(self):
        if getattr(self, "buf", None) is not None:
            self.buf.close()
            self.buf = None
        if self._fh:
            self._fh.close()
            self._fh = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __getitem__(self, path):
        return self.root[path] if path.strip("/") else self.root

    def keys(self):
        return self.root.keys()

    # -- primitives
    def _addr(self, pos):
        v = int.from_bytes(self.buf[pos:pos + self.osz], "little")
        return UNDEF if v == (1 << (8 * self.osz)) - 1 else v

    def _len(self, pos):
        return int.from_bytes(self.buf[pos:pos + self.lsz], "little")

    def _unpack_addrs(self, pos, n):
        return [self._addr(pos + i * self.osz) for i in range(n)]

    def _check(self, pos, n, what):
        if pos < 0 or pos + n > len(self.buf):
            raise H5Error(f"{self.path}: {what} at {pos}+{n} runs past the end of the file (truncated?)")

    # -- object headers
    def _read_header(self, addr):
        b = self.buf
        self._check(addr, 16, "object header")
        msgs = []
        if b[addr:addr + 4] == b"OHDR":                            # version 2
            flags = b[addr + 5]
            p = addr + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            nb = 1 << (flags & 3)
            size0 = int.from_bytes(b[p:p + nb], "little")
            p += nb
            blocks = [(p, p + size0)]
            track = bool(flags & 4)
            while blocks:
                p, end = blocks.pop(0)
                while p + 4 + (2 if track else 0) <= end:
                    t, size, fl = b[p], struct.unpack_from("<H", b, p + 1)[0], b[p + 3]
                    p += 4 + (2 if track else 0)
                    if t == 0x10:
                        off, ln = self._addr(p), self._len(p + self.osz)
                        self._check(off + self.base, ln, "object header continuation")
                        blocks.append((off + self.base + 4, off + self.base + ln - 4))      # "OCHK" ... checksum
                    elif t != 0:
                        msgs.append((t, fl, p, size))
                    p += size
            return msgs
        if b[addr] != 1:
            raise H5Error(f"{self.path}: object header version {b[addr]} at {addr} is not supported")
        nmsg, _, hsize = struct.unpack_from("<HII", b, addr + 2)
        blocks = [(addr + 16, addr + 16 + hsize)]
        seen = 0
        while blocks and seen < nmsg:
            p, end = blocks.pop(0)
            self._check(p, end - p, "object header block")
            while p + 8 <= end and seen < nmsg:
                t, size, fl = struct.unpack_from("<HHB", b, p)
                p += 8
                seen += 1
                if t == 0x10:
                    off, ln = self._addr(p), self._len(p + self.osz)
                    blocks.append((off + self.base, off + self.base + ln))
                elif t != 0:
                    msgs.append((t, fl, p, size))
                p += size
        return msgs

    def _parse_dataspace(self, pos):
        b = self.buf
        ver, rank, flags = b[pos], b[pos + 1], b[pos + 2]
        if ver == 1:
            p = pos + 8
        elif ver == 2:
            if b[pos + 3] == 2:                                    # null dataspace
                return None, None
            p = pos + 4
        else:
            raise H5Error(f"dataspace message version {ver} is not supported")
        dims = tuple(self._len(p + i * self.lsz) for i in range(rank))
        p += rank * self.lsz
        mx = dims
        if flags & 1:
            mx = tuple(None if (v := self._len(p + i * self.lsz)) == (1 << (8 * self.lsz)) - 1 else v for i in range(rank))
        return dims, mx

    def _parse_filters(self, m):
        if m is None:
            return []
        b, p = self.buf, m[2]
        ver, n = b[p], b[p + 1]
        p += 8 if ver == 1 else 2
        out = []
        for _ in range(n):
            fid = struct.unpack_from("<H", b, p)[0]
            p += 2
            nlen = 0
            if ver == 1 or fid >= 256:
                nlen = struct.unpack_from("<H", b, p)[0]
                p += 2
            fl, nv = struct.unpack_from("<HH", b, p)
            p += 4
            if ver == 1:
                nlen = (nlen + 7) & ~7
            p += nlen
            vals = struct.unpack_from(f"<{nv}I", b, p)
            p += 4 * nv
            if ver == 1 and nv & 1:
                p += 4
            out.append((fid, vals))
        return out

    # -- data
    def _unfilter(self, raw, filters, mask, esz, name):
        for i in range(len(filters) - 1, -1, -1):
            if mask & (1 << i):
                continue
            fid, vals = filters[i]
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:
                k = vals[0] if vals else esz
                a = np.frombuffer(raw, np.uint8)
                n = a.size // k
                raw = a[:n * k].reshape(k, n).T.tobytes() + a[n * k:].tobytes()
            elif fid == 3:
                raw = raw[:-4]
            else:
                names = {4: "szip", 305: "lzo", 307: "bzip2", 32001: "blosc", 32026: "blosc2"}
                raise H5Error(f"{name}: filter {names.get(fid, fid)} is not supported (deflate / shuffle / fletcher32 are)")
        return raw

    def _read_layout(self, pos, shape, esz, filters, name):
        b = self.buf
        ver = b[pos]
        if ver == 3:
            cls = b[pos + 1]
            if cls == 0:
                size = struct.unpack_from("<H", b, pos + 2)[0]
                return bytes(b[pos + 4:pos + 4 + size])
            if cls == 1:
                addr, size = self._addr(pos + 2), self._len(pos + 2 + self.osz)
                if addr == UNDEF:
                    return None
                self._check(addr + self.base, size, f"{name} data")
                return bytes(b[addr + self.base:addr + self.base + size])
            if cls == 2:
                nd = b[pos + 2]
                bt = self._addr(pos + 3)
                cdims = struct.unpack_from(f"<{nd}I", b, pos + 3 + self.osz)
                return self._read_chunked(bt, cdims, shape, esz, filters, name)
            raise H5Error(f"{name}: layout class {cls} is not supported")
        if ver in (1, 2):
            nd, cls = b[pos + 1], b[pos + 2]
            p = pos + 8
            addr = UNDEF
            if cls != 0:
                addr = self._addr(p)
                p += self.osz
            dims = struct.unpack_from(f"<{nd}I", b, p)
            p += 4 * nd
            if cls == 0:
                size = struct.unpack_from("<I", b, p)[0]
                return bytes(b[p + 4:p + 4 + size])
            if cls == 1:
                if addr == UNDEF:
                    return None
                size = int(np.prod(dims, dtype=np.int64)) * esz
                return bytes(b[addr + self.base:addr + self.base + size])
            return self._read_chunked(addr, dims + (struct.unpack_from("<I", b, p)[0],), shape, esz, filters, name)
        raise H5Error(f"{name}: layout message version {ver} is not supported")

    def _read_chunked(self, bt, cdims, shape, esz, filters, name):
        nd = len(cdims) - 1                                        # the last "dimension" is the element size
        cshape = tuple(int(c) for c in cdims[:nd])
        out = np.zeros(tuple(shape) + (esz,), np.uint8)
        if bt == UNDEF or out.size == 0:
            return out.tobytes()
        csize = int(np.prod(cshape, dtype=np.int64)) * esz
        for offs, size, mask, addr in self._walk_chunk_btree(bt + self.base, nd):
            self._check(addr + self.base, size, f"{name} chunk")
            raw = self._unfilter(bytes(self.buf[addr + self.base:addr + self.base + size]), filters, mask, esz, name)
            if len(raw) < csize:
                raise H5Error(f"{name}: chunk at {offs} holds {len(raw)} bytes, expected {csize}")
            c = np.frombuffer(raw, np.uint8, csize).reshape(cshape + (esz,))
            sl_o = tuple(slice(o, min(o + c_, s)) for o, c_, s in zip(offs, cshape, shape))
            sl_c = tuple(slice(0, s.stop - s.start) for s in sl_o)
            if all(s.stop > s.start for s in sl_o):
                out[sl_o] = c[sl_c]
        return out.tobytes()

    def _walk_chunk_btree(self, addr, nd):
        b = self.buf
        self._check(addr, 24, "chunk B-tree node")
        if b[addr:addr + 4] != b"TREE" or b[addr + 4] != 1:
            raise H5Error(f"{self.path}: bad chunk B-tree node at {addr}")
        level, used = b[addr + 5], struct.unpack_from("<H", b, addr + 6)[0]
        p = addr + 8 + 2 * self.osz
        ksz = 8 + 8 * (nd + 1)
        for _ in range(used):
            size, mask = struct.unpack_from("<II", b, p)
            offs = struct.unpack_from(f"<{nd}Q", b, p + 8)
            child = self._addr(p + ksz)
            if level == 0:
                yield offs, size, mask, child
            else:
                yield from self._walk_chunk_btree(child + self.base, nd)
            p += ksz + self.osz

    def _walk_group_btree(self, bt, heap):
        b = self.buf
        hp = heap + self.base
        self._check(hp, 8 + 2 * self.lsz + self.osz, "local heap")
        if b[hp:hp + 4] != b"HEAP":
            raise H5Error(f"{self.path}: bad local heap at {hp}")
        data = self._addr(hp + 8 + 2 * self.lsz) + self.base
        stack = [bt + self.base]
        while stack:
            a = stack.pop()
            self._check(a, 8, "group B-tree node")
            if b[a:a + 4] == b"TREE":
                if b[a + 4] != 0:
                    raise H5Error(f"{self.path}: B-tree node at {a} is not a group node")
                used = struct.unpack_from("<H", b, a + 6)[0]
                p = a + 8 + 2 * self.osz + self.lsz
                for _ in range(used):
                    stack.append(self._addr(p) + self.base)
                    p += self.osz + self.lsz
            elif b[a:a + 4] == b"SNOD":
                n = struct.unpack_from("<H", b, a + 6)[0]
                p = a + 8
                for _ in range(n):
                    noff, hdr = self._addr(p), self._addr(p + self.osz)
                    e = b.find(b"\0", data + noff)
                    yield bytes(b[data + noff:e]).decode(), hdr + self.base
                    p += 2 * self.osz + 24
            else:
                raise H5Error(f"{self.path}: neither TREE nor SNOD at {a}")

    def _heap_object(self, addr, idx):
        col = self._gcol.get(addr)
        if col is None:
            b = self.buf
            a = addr + self.base
            self._check(a, 16, "global heap collection")
            if b[a:a + 4] != b"GCOL":
                raise H5Error(f"{self.path}: bad global heap collection at {a}")
            size = self._len(a + 8)
            col = {}
            p, end = a + 8 + self.lsz, a + size
            while p + 8 + self.lsz <= end:
                i = struct.unpack_from("<H", b, p)[0]
                osize = self._len(p + 8)
                if i == 0:
                    break
                col[i] = (p + 8 + self.lsz, osize)
                p += 8 + self.lsz + ((osize + 7) & ~7)
            self._gcol[addr] = col
        if idx not in col:
            raise H5Error(f"{self.path}: global heap object {idx} missing from the collection at {addr}")
        p, n = col[idx]
        return bytes(self.buf[p:p + n])

    def _decode(self, t, shape, raw):
        if shape is None:                                          # null dataspace
            return "" if t.kind in ("str", "vlen_str") else None
        n = int(np.prod(shape, dtype=np.int64))
        if t.kind in ("vlen", "vlen_str"):
            out = np.empty(n, object)
            step = 4 + self.osz + 4
            for i in range(n):
                ln = struct.unpack_from("<I", raw, i * step)[0]
                addr = int.from_bytes(raw[i * step + 4:i * step + 4 + self.osz], "little")
                idx = struct.unpack_from("<I", raw, i * step + 4 + self.osz)[0]
                data = b"" if ln == 0 or addr == 0 else self._heap_object(addr, idx)
                if t.kind == "vlen_str":
                    out[i] = data[:ln].decode("utf-8", "replace")
                else:
                    out[i] = np.frombuffer(data, t.base.dtype, ln)
            return out.reshape(shape)
        return np.frombuffer(raw, t.dtype, n).reshape(shape).copy()

    def _read_attribute(self, pos, size):
        b = self.buf
        ver = b[pos]
        nsz, tsz, ssz = struct.unpack_from("<HHH", b, pos + 2)
        p = pos + 8 + (1 if ver == 3 else 0)
        pad = (lambda x: (x + 7) & ~7) if ver == 1 else (lambda x: x)
        name = bytes(b[p:p + nsz]).split(b"\0")[0].decode()
        p += pad(nsz)
        if ver >= 2 and b[pos + 1] & 3:
            raise H5Error(f"attribute {name!r}: shared datatype / dataspace messages are not supported")
        t, _ = _parse_type(b, p, self.osz)
        p += pad(tsz)
        shape, _ = self._parse_dataspace(p)
        p += pad(ssz)
        n = 0 if shape is None else int(np.prod(shape, dtype=np.int64))
        v = self._decode(t, shape, bytes(b[p:p + n * t.size]))
        if shape == ():
            v = v[()]
            if t.kind == "str":
                v = bytes(v).split(b"\0")[0].decode("utf-8", "replace")
            elif isinstance(v, np.generic):
                v = v.item()
        return name, v


# ------------------------------------------------------------------------------------------------ pandas layer
class Frame(OrderedDict):
    """Columns of a pandas frame (name -> 1-D numpy array) in their file order, plus `.index` (array, or a list of
    arrays for a MultiIndex) and `.index_names`."""

    def __init__(self, cols=(), index=None, index_names=None, series=False):
        super().__init__(cols)
        self.index = index
        self.index_names = index_names
        self.series = series              # one column, stored / to be stored as a pandas Series (`pandas_type` = series)

    @classmethod
    def from_pandas(cls, obj):
        """DataFrame / Series -> Frame (object columns stay object arrays; a default RangeIndex becomes None)."""
        import pandas as pd
        series = isinstance(obj, pd.Series)
        df = obj.to_frame(obj.name if obj.name is not None else "values") if series else obj
        idx = df.index
        if isinstance(idx, pd.MultiIndex):
            index, names = [idx.get_level_values(i).to_numpy() for i in range(idx.nlevels)], list(idx.names)
        elif isinstance(idx, pd.RangeIndex) and idx.start == 0 and idx.step == 1 and idx.name is None:
            index, names = None, None
        else:
            index, names = idx.to_numpy(), [idx.name]
        fr = cls([(str(c), df[c].to_numpy()) for c in df.columns], index=index, index_names=names, series=series)
        if series and obj.name is None:
            fr.series_name = None
        return fr

    @property
    def n_rows(self):
        for v in self.values():
            return len(v)
        if self.index is None:
            return 0
        return len(self.index[0]) if isinstance(self.index, list) else len(self.index)

    def to_pandas(self):
        import pandas as pd
        idx = None
        if isinstance(self.index, list):
            idx = pd.MultiIndex.from_arrays(self.index, names=self.index_names)
        elif self.index is not None:
            idx = pd.Index(self.index, name=(self.index_names or [None])[0])
        if self.series and len(self) == 1:
            (name, vals), = self.items()
            return pd.Series(vals, index=idx, name=getattr(self, "series_name", name))
        return pd.DataFrame({k: v for k, v in self.items()}, index=idx, columns=list(self.keys()))


# ---- unpickling what PyTables / pandas pickled: only data constructors, and array states that add up -------------
class _CheckedArray(np.ndarray):
    """ndarray whose pickle state is validated before numpy installs it: a damaged object block must raise, not read
    past the end of a too-short object list (which numpy does not check for object dtypes)."""

    def __setstate__(self, state):
        try:
            _, shape, dtype, _, raw = state
            n = 1
            for d in shape:
                n *= int(d)
            ok = (isinstance(raw, list) and len(raw) == n) if dtype.hasobject else \
                (isinstance(raw, (bytes, bytearray)) and len(raw) == n * dtype.itemsize)
        except Exception:
            ok = False
        if not ok:
            raise pickle.UnpicklingError("ndarray state does not match its shape (damaged object block)")
        super().__setstate__(state)


def _checked_reconstruct(subtype, shape, typecode):
    return np.ndarray.__new__(_CheckedArray, shape, typecode)


try:
    from numpy._core.multiarray import scalar as _np_scalar
except ImportError:                                               # numpy 1.x
    from numpy.core.multiarray import scalar as _np_scalar

class _NdarrayToken:
    """Stands for `numpy.ndarray` where pickles name it: as the `subtype` argument of `_reconstruct` (ignored there -
    the array is always built as a _CheckedArray).  Calling it is refused: `ndarray(shape, dtype('O'))` followed by a
    BUILD would reach numpy's unchecked `__setstate__`, which reads past a short object list (a crash, not an error)."""

    def __call__(self, *args, **kwargs):
        raise pickle.UnpicklingError("numpy.ndarray may only appear as the array type of a reconstructed array")


_SAFE_GLOBALS = {
    ("numpy", "ndarray"): _NdarrayToken(), ("numpy", "dtype"): np.dtype,
    ("builtins", "tuple"): tuple, ("builtins", "list"): list, ("builtins", "dict"): dict, ("builtins", "set"): set,
    ("builtins", "frozenset"): frozenset, ("builtins", "complex"): complex, ("builtins", "bytearray"): bytearray,
    ("builtins", "slice"): slice, ("builtins", "range"): range, ("__builtin__", "tuple"): tuple, ("__builtin__", "list"): list,
    ("__builtin__", "dict"): dict, ("__builtin__", "set"): set, ("__builtin__", "frozenset"): frozenset,
    ("__builtin__", "complex"): complex, ("__builtin__", "unicode"): str, ("__builtin__", "long"): int,
    ("collections", "OrderedDict"): OrderedDict, ("_codecs", "encode"): __import__("_codecs").encode,
}
for _m in ("numpy.core.multiarray", "numpy._core.multiarray"):
    _SAFE_GLOBALS[(_m, "_reconstruct")] = _checked_reconstruct
    _SAFE_GLOBALS[(_m, "scalar")] = _np_scalar
for _n in ("datetime", "date", "time", "timedelta", "timezone"):
    _SAFE_GLOBALS[("datetime", _n)] = getattr(__import__("datetime"), _n)
_SAFE_GLOBALS[("decimal", "Decimal")] = __import__("decimal").Decimal


class _DataUnpickler(pickle.Unpickler):
    """Unpickler for object blocks / PyTables attributes: data constructors only.  Reading a table must not run
    whatever callable a file names - pandas itself offers no such guard."""

    def find_class(self, module, name):
        obj = _SAFE_GLOBALS.get((module, name))
        if obj is None:
            raise pickle.UnpicklingError(f"global {module}.{name} is not allowed in an HDF5 object block")
        return obj


def _loads(raw: bytes):
    import io
    out = _DataUnpickler(io.BytesIO(raw)).load()

    def plain(x):
        if isinstance(x, _CheckedArray):
            x = x.view(np.ndarray)
            if x.dtype == object:
                flat = x.reshape(-1)
                for i, y in enumerate(flat):
                    if isinstance(y, _CheckedArray):
                        flat[i] = plain(y)
        return x
    return plain(out)


def _unpickle_attr(v):
    """PyTables keeps non-string Python attribute values as pickles inside string attributes."""
    if isinstance(v, str):
        raw = v.encode("latin-1", "ignore")
    elif isinstance(v, (bytes, np.bytes_)):
        raw = bytes(v)
    else:
        return v
    if raw.endswith(b"."):
        try:
            return _loads(raw)
        except Exception:
            return v
    return v


def _read_array(group, key):
    """pandas.io.pytables.GenericFixed.read_array on our reader."""
    ds = group[key]
    at = ds.attrs
    transposed = bool(at.get("transposed", False))
    if at.get("CLASS") == "VLARRAY":
        rows = ds.read()
        if rows.size == 0:
            ret = np.empty(0, object)
        else:
            try:
                ret = _loads(rows.reshape(-1)[0].tobytes())
            except H5Error:
                raise
            except Exception as e:                                 # damaged or hostile pickle
                raise H5Error(f"{ds.name}: object block cannot be unpickled ({type(e).__name__}: {e})")
    else:
        ret = ds.read()
        shape = _unpickle_attr(at.get("shape"))
        vt = at.get("value_type")
        if shape is not None and not isinstance(shape, str):
            ret = np.empty(tuple(shape), dtype=np.dtype(vt) if isinstance(vt, str) and vt else ret.dtype)
        elif isinstance(vt, str) and vt.startswith("datetime64"):
            ret = ret.view("M8[ns]")
        elif vt == "timedelta64":
            ret = ret.view("m8[ns]")
    if transposed:
        ret = ret.T
    return ret


def _read_index_node(group, key, encoding="UTF-8"):
    ds = group[key]
    kind = ds.attrs.get("kind")
    data = _read_array(group, key)
    if kind == "string":
        data = np.array([x.split(b"\0")[0].decode(encoding or "UTF-8") for x in np.asarray(data).reshape(-1)], dtype=object)
    elif kind in ("datetime64", "datetime"):
        data = np.asarray(data).view("M8[ns]")
    elif kind == "timedelta64":
        data = np.asarray(data).view("m8[ns]")
    name = _unpickle_attr(ds.attrs.get("name"))
    return data, (None if isinstance(name, str) and name == "N." else name)


def _read_axis(group, key):
    variety = group.attrs.get(f"{key}_variety", "regular")
    enc = group.attrs.get("encoding", "UTF-8")
    if variety == "regular":
        data, name = _read_index_node(group, key, enc)
        return data, [name]
    if variety == "multi":
        n = int(group.attrs[f"{key}_nlevels"])
        arrays, names = [], []
        for i in range(n):
            lev, lev_name = _read_index_node(group, f"{key}_level{i}", enc)
            codes = np.asarray(_read_array(group, f"{key}_label{i}"))
            col = np.asarray(lev, dtype=object if lev.dtype == object else lev.dtype)[np.where(codes < 0, 0, codes)] if len(lev) else np.empty(len(codes), object)
            if (codes < 0).any():
                col = col.astype(object)
                col[codes < 0] = None
            arrays.append(col)
            names.append(lev_name)                                # pandas keeps a level's name on the level node
        return arrays, names
    raise H5Error(f"axis variety {variety!r} is not supported")


def _decode_strings(a, encoding, nan_rep):
    out = np.empty(a.shape, object)
    flat = out.reshape(-1)
    for i, x in enumerate(a.reshape(-1)):
        t = bytes(x).split(b"\0")[0].decode(encoding or "UTF-8")
        flat[i] = np.nan if t == nan_rep else t
    return out


def _read_frame_table(g, key):
    """pandas format="table" (`frame_table`): one PyTables Table of records {index, values_block_<i>[k], data
    columns...}; `<field>_kind` on the table lists the frame columns a field carries, `non_index_axes` on the group
    the column order.  Strings are fixed-width bytes with `nan_rep` standing for missing."""
    if int(g.attrs.get("levels", 1)) != 1:
        raise H5Error(f"{key}: table-format frames with a MultiIndex are not supported")
    t = g["table"]
    rec = t.read()
    enc, nan_rep = g.attrs.get("encoding", "UTF-8"), g.attrs.get("nan_rep", "nan")
    axes = _unpickle_attr(g.attrs.get("non_index_axes"))
    order = list(axes[0][1]) if axes else None
    fields = _unpickle_attr(g.attrs.get("values_cols"))
    cols = {}
    for fld in fields:
        names = _unpickle_attr(t.attrs.get(f"{fld}_kind"))
        dts = t.attrs.get(f"{fld}_dtype", "")
        a = rec[fld]
        if a.ndim == 1:
            a = a[:, None]
        if a.shape[1] != len(names):
            raise H5Error(f"{key}: field {fld} is {a.shape[1]} wide for the columns {names}")
        for j, nm in enumerate(names):
            v = np.ascontiguousarray(a[:, j])
            if v.dtype.kind == "S":
                v = _decode_strings(v, enc, nan_rep)
            elif isinstance(dts, str) and dts.startswith("datetime64"):
                v = v.view("M8[ns]")
            elif isinstance(dts, str) and dts.startswith("timedelta64"):
                v = v.view("m8[ns]")
            cols[nm] = v
    index = rec["index"] if "index" in rec.dtype.names else None
    if index is not None and index.dtype.kind == "S":
        index = _decode_strings(index, enc, nan_rep)
    order = order if order is not None else list(cols)
    return Frame([(c, cols[c]) for c in order], index=None if index is None else np.ascontiguousarray(index), index_names=[None])


def read_hdf(path, key=None):
    """pandas.read_hdf for fixed-format frames and series and for table-format frames.  Returns a Frame (a series comes back as a one-column
    frame named by the series' name, or "values")."""
    with H5File(path) as f:
        if key is None:
            cands = [k for k in f.keys() if "pandas_type" in f[k].attrs]
            if len(cands) != 1:
                raise ValueError(f"key must be provided when the file holds {len(cands)} pandas objects: {cands}")
            key = cands[0]
        try:
            g = f[key]
        except KeyError:
            raise KeyError(f"No object named {key} in the file")
        ptype = g.attrs.get("pandas_type")
        if ptype == "frame":
            cols, _ = _read_axis(g, "axis0")
            index, names = _read_axis(g, "axis1")
            got = {}
            for i in range(int(g.attrs["nblocks"])):
                items, _ = _read_axis(g, f"block{i}_items")
                vals = np.asarray(_read_array(g, f"block{i}_values"))
                if vals.ndim != 2 or vals.shape[0] != len(items):
                    raise H5Error(f"{key}/block{i}_values has shape {vals.shape} for {len(items)} columns")
                for j, c in enumerate(items):
                    got[c] = np.ascontiguousarray(vals[j])
            missing = [c for c in cols if c not in got]
            if missing:
                raise H5Error(f"{key}: columns {missing} are in axis0 but in no block")
            return Frame([(c, got[c]) for c in cols], index=index, index_names=names)
        if ptype == "series":
            index, names = _read_axis(g, "index")
            vals = np.asarray(_read_array(g, "values"))
            name = _unpickle_attr(g.attrs.get("name"))
            fr = Frame([("values" if name is None or name == "N." else name, vals)], index=index, index_names=names, series=True)
            if name is None or name == "N.":
                fr.series_name = None
            return fr
        if ptype == "frame_table":
            return _read_frame_table(g, key)
        if ptype in ("series_table", "appendable_frame", "appendable_series", "appendable_multiframe"):
            raise H5Error(f"{key}: {ptype} stores are not supported (fixed-format frames / series and frame_table are)")
        raise H5Error(f"{key}: not a pandas object (pandas_type = {ptype!r})")


# ------------------------------------------------------------------------------------------------ writer
def _pad8(b):
    return b + bytes(-len(b) % 8)


def _enc_type(dt, cset=0):
    """Datatype message of a numpy dtype (version 1 encodings, little endian - what PyTables writes)."""
    dt = np.dtype(dt)
    if dt.kind in "iu":
        return struct.pack("<BBBBIHH", 0x10, 0x08 if dt.kind == "i" else 0, 0, 0, dt.itemsize, 0, 8 * dt.itemsize)
    if dt.kind == "f":
        exp_loc, exp_size, man_size, bias = {2: (10, 5, 10, 15), 4: (23, 8, 23, 127), 8: (52, 11, 52, 1023)}[dt.itemsize]
        return struct.pack("<BBBBIHHBBBBI", 0x11, 0x20, 8 * dt.itemsize - 1, 0, dt.itemsize, 0, 8 * dt.itemsize,
                           exp_loc, exp_size, 0, man_size, bias)
    if dt.kind == "b":
        return struct.pack("<BBBBIHH", 0x14, 0, 0, 0, 1, 0, 8)
    if dt.kind == "S":
        return struct.pack("<BBBBI", 0x13, cset << 4, 0, 0, max(1, dt.itemsize))
    raise H5Error(f"dtype {dt} cannot be stored")


_VLEN_U8 = struct.pack("<BBBBI", 0x19, 0, 0, 0, 16) + _enc_type(np.uint8)


def _enc_space(shape, unlimited=False):
    if shape is None:
        return bytes([2, 0, 0, 2])                                  # null dataspace (version 2)
    b = struct.pack("<BBB5x", 1, len(shape), 1 if unlimited else 0) + b"".join(struct.pack("<Q", d) for d in shape)
    if unlimited:
        b += struct.pack("<Q", UNDEF) * len(shape)
    return b


class _Image:
    """File image under construction: append-only allocation, 8-byte aligned."""

    def __init__(self):
        self.b = bytearray(96)                                      # the version-0 superblock goes here last
        self.heap_objs = []                                         # global heap: payloads of variable-length rows

    def put(self, data):
        self.b += bytes(-len(self.b) % 8)
        addr = len(self.b)
        self.b += data
        return addr

    def header(self, msgs):
        """Version-1 object header of (type, flags, payload) messages in one block."""
        body = b"".join(struct.pack("<HHB3x", t, len(_pad8(d)), fl) + _pad8(d) for t, fl, d in msgs)
        return self.put(struct.pack("<BBHII4x", 1, 0, len(msgs), 1, len(body)) + body)

    @staticmethod
    def attr(name, value):
        nm = name.encode() + b"\0"
        if isinstance(value, _Pickled):                             # PyTables: pickle (protocol 0) in an ASCII string
            raw = value.raw
            dt, sp, data = struct.pack("<BBBBI", 0x13, 0, 0, 0, len(raw)), _enc_space(()), raw
        elif isinstance(value, str):
            raw = value.encode("utf-8")
            if raw:
                dt, sp, data = struct.pack("<BBBBI", 0x13, 0x10, 0, 0, len(raw)), _enc_space(()), raw
            else:
                dt, sp, data = struct.pack("<BBBBI", 0x13, 0x10, 0, 0, 1), _enc_space(None), b""
        elif isinstance(value, (bool, np.bool_)):
            dt, sp, data = _enc_type(np.bool_), _enc_space(()), bytes([1 if value else 0])
        elif isinstance(value, (int, np.integer)):
            dt, sp, data = _enc_type(np.int64), _enc_space(()), struct.pack("<q", int(value))
        elif isinstance(value, (float, np.floating)):
            dt, sp, data = _enc_type(np.float64), _enc_space(()), struct.pack("<d", float(value))
        else:
            raise H5Error(f"attribute {name!r}: values of type {type(value).__name__} cannot be stored")
        return (0x000C, 0, struct.pack("<BBHHH", 1, 0, len(nm), len(dt), len(sp)) + _pad8(nm) + _pad8(dt) + _pad8(sp) + data)

    def dataset(self, arr, attrs):
        """Contiguous dataset of a numeric / bool / fixed-string array."""
        arr = np.ascontiguousarray(arr)
        if arr.dtype.byteorder == ">":
            arr = arr.astype(arr.dtype.newbyteorder("<"))
        raw = arr.tobytes()
        addr = self.put(raw) if raw else UNDEF
        msgs = [(0x0001, 0, _enc_space(arr.shape)), (0x0003, 1, _enc_type(arr.dtype)),
                (0x0005, 1, bytes([2, 2, 2, 1, 0, 0, 0, 0])),
                (0x0008, 0, struct.pack("<BBQQ", 3, 1, addr, len(raw)))]
        return self.header(msgs + [self.attr(k, v) for k, v in attrs])

    def vlarray(self, payload, attrs):
        """PyTables VLArray holding one row of bytes (pandas' pickled object block): an extendible 1-D dataset of
        variable-length uint8 with a one-chunk B-tree."""
        self.heap_objs.append(payload)
        idx = len(self.heap_objs)
        elem = self.put(struct.pack("<IQI", len(payload), 0, idx))  # heap address patched in finish()
        self._vl_fix.append(elem)
        key = 8 + 8 * 2
        node = bytearray(24 + 64 * 8 + 65 * key)                    # full node for K = 32 (the library reads it whole)
        struct.pack_into("<4sBBHQQ", node, 0, b"TREE", 1, 0, 1, UNDEF, UNDEF)
        struct.pack_into("<IIQQQ", node, 24, 16, 0, 0, 0, elem)
        struct.pack_into("<IIQQ", node, 24 + key + 8, 0, 0, 1, 16)
        bt = self.put(bytes(node))
        msgs = [(0x0001, 0, _enc_space((1,), unlimited=True)), (0x0003, 1, _VLEN_U8),
                (0x0005, 1, bytes([2, 3, 0, 1, 0, 0, 0, 0])),
                (0x0008, 0, struct.pack("<BBBQII", 3, 2, 2, bt, 1, 16))]
        return self.header(msgs + [self.attr(k, v) for k, v in attrs])

    _vl_fix = None

    def group(self, children, attrs, leaf_k):
        """Symbol-table group of {name: (header address, None | (btree, heap))}; returns (header, btree, heap)."""
        names = sorted(children)
        seg = bytearray(8)
        offs = {}
        for n in names:
            offs[n] = len(seg)
            seg += _pad8(n.encode() + b"\0")
        data = self.put(bytes(seg))
        heap = self.put(struct.pack("<4sB3xQQQ", b"HEAP", 0, len(seg), 1, data))
        snod = bytearray(8 + 2 * leaf_k * 40)
        struct.pack_into("<4sBBH", snod, 0, b"SNOD", 1, 0, len(names))
        for i, n in enumerate(names):
            hdr, sub = children[n]
            if sub is None:
                struct.pack_into("<QQII16x", snod, 8 + 40 * i, offs[n], hdr, 0, 0)
            else:
                struct.pack_into("<QQIIQQ", snod, 8 + 40 * i, offs[n], hdr, 1, 0, sub[0], sub[1])
        node = bytearray(24 + 32 * 8 + 33 * 8)                      # internal K = 16
        struct.pack_into("<4sBBHQQ", node, 0, b"TREE", 0, 0, 1 if names else 0, UNDEF, UNDEF)
        if names:
            struct.pack_into("<QQQ", node, 24, 0, self.put(bytes(snod)), offs[names[-1]])
        bt = self.put(bytes(node))
        hdr = self.header([(0x0011, 0, struct.pack("<QQ", bt, heap))] + [self.attr(k, v) for k, v in attrs])
        return hdr, bt, heap

    def finish(self, root, leaf_k):
        hdr, bt, heap = root
        if self.heap_objs:
            body = bytearray()
            for i, p in enumerate(self.heap_objs, 1):
                body += struct.pack("<HHIQ", i, 1, 0, len(p)) + _pad8(p)
            size = max(4096, (16 + len(body) + 16 + 4095) & ~4095)
            free = size - 16 - len(body)
            body += struct.pack("<HHIQ", 0, 0, 0, free) + bytes(free - 16)
            gcol = self.put(struct.pack("<4sB3xQ", b"GCOL", 1, size) + bytes(body))
            for e in self._vl_fix:
                struct.pack_into("<Q", self.b, e + 4, gcol)
        self.b += bytes(-len(self.b) % 8)
        sb = SIGNATURE + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + struct.pack("<HHI", leaf_k, 16, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, len(self.b), UNDEF)
        sb += struct.pack("<QQIIQQ", 0, hdr, 1, 0, bt, heap)
        self.b[:96] = sb
        return bytes(self.b)


class _Pickled:
    def __init__(self, obj):
        self.raw = pickle.dumps(obj, protocol=0)


_ARRAY_ATTRS = [("CLASS", "ARRAY"), ("VERSION", "2.4"), ("TITLE", ""), ("FLAVOR", "numpy")]
_VLARRAY_ATTRS = [("CLASS", "VLARRAY"), ("VERSION", "1.4"), ("TITLE", ""), ("PSEUDOATOM", "object")]
_GROUP_ATTRS = [("CLASS", "GROUP"), ("VERSION", "1.0"), ("TITLE", "")]


def _write_array(img, value, extra=()):
    """pandas.io.pytables.GenericFixed.write_array: (header address) of `value` stored the way pandas stores it."""
    value = np.asarray(value)
    empty = value.size == 0
    transposed = False
    attrs = []
    if empty:                                                      # pandas: a dummy element + the real shape and dtype
        attrs = [("value_type", str(value.dtype)), ("shape", _Pickled(tuple(int(x) for x in value.shape)))]
        stored = np.empty((1,) * value.ndim, np.float64)
        stored[...] = 0
        return img.dataset(stored, _ARRAY_ATTRS + [("transposed", False)] + attrs + list(extra))
    value = value.T
    transposed = True
    if value.dtype == object:
        return img.vlarray(pickle.dumps(value, protocol=4), _VLARRAY_ATTRS + [("transposed", transposed)] + list(extra))
    if value.dtype.kind == "M":
        return img.dataset(value.view("i8"), _ARRAY_ATTRS + [("transposed", transposed), ("value_type", "datetime64")] + list(extra))
    if value.dtype.kind == "m":
        return img.dataset(value.view("i8"), _ARRAY_ATTRS + [("transposed", transposed), ("value_type", "timedelta64")] + list(extra))
    if value.dtype.kind == "U":
        raise H5Error("unicode arrays must be converted to object or bytes first")
    return img.dataset(value, _ARRAY_ATTRS + [("transposed", transposed)] + list(extra))


def _write_index(img, children, key, values, name=None, extra=()):
    """One index node: integers as int64, floats as float64, strings as a fixed-width byte array (kind "string"),
    anything else as a pickled object array.  The `name` attribute is the string itself, or a pickle for None - what
    PyTables stores for a str / a non-str Python attribute."""
    values = np.asarray(values)
    nm = ("name", name if isinstance(name, str) and name else _Pickled(name))
    if values.dtype.kind in "iu":
        children[key] = (_write_array(img, values.astype(np.int64), [("kind", "integer"), nm] + list(extra)), None)
    elif values.dtype.kind == "f":
        children[key] = (_write_array(img, values.astype(np.float64), [("kind", "float"), nm] + list(extra)), None)
    elif values.dtype.kind in "OSU" and all(isinstance(x, (str, bytes)) for x in values.reshape(-1)):
        enc = [x.encode("utf-8") if isinstance(x, str) else x for x in values.reshape(-1)]
        width = max([len(x) for x in enc] + [1])
        children[key] = (_write_array(img, np.array(enc, dtype=f"S{width}"), [("kind", "string"), nm] + list(extra)), None)
    else:
        children[key] = (_write_array(img, values.astype(object), [("kind", "object"), nm] + list(extra)), None)


def _write_axis(img, children, gattrs, key, index, names, n):
    if isinstance(index, list):                                    # MultiIndex: levels + integer codes
        gattrs += [(f"{key}_variety", "multi"), (f"{key}_nlevels", len(index))]
        names = list(names or [None] * len(index))
        for i, col in enumerate(index):
            col = np.asarray(col)
            lev, codes = np.unique(col, return_inverse=True) if len(col) else (col, np.zeros(0, np.int64))
            # pandas keeps a level's name on the level node, twice: `name`, and `<axis>_name<name>` (sic)
            lvl_name = names[i] if isinstance(names[i], str) and names[i] else _Pickled(names[i])
            _write_index(img, children, f"{key}_level{i}", lev, names[i], extra=[(f"{key}_name{names[i]}", lvl_name)])
            cdt = np.int8 if len(lev) < 128 else (np.int16 if len(lev) < 32768 else (np.int32 if len(lev) < 2 ** 31 else np.int64))
            children[f"{key}_label{i}"] = (_write_array(img, codes.astype(cdt)), None)
        return
    gattrs.append((f"{key}_variety", "regular"))
    if index is None:
        index = np.arange(n, dtype=np.int64)
    _write_index(img, children, key, index, (names or [None])[0])


def write_hdf(path, objects, mode="w"):
    """Write {key: Frame | dict of equal-length 1-D arrays} as pandas fixed-format frames (`DataFrame.to_hdf(path,
    key)` for every key).  mode "w" replaces the file; mode "a" (pandas' default) keeps the frames already stored
    under other keys - the file is small bookkeeping next to the data, so it is simply rewritten.  Object columns are
    pickled exactly as pandas does."""
    if mode not in ("w", "a"):
        raise ValueError(f"mode must be 'w' or 'a', not {mode!r}")
    if mode == "a":
        import os
        if os.path.exists(path) and os.path.getsize(path):
            with H5File(path) as f:
                old = [k for k in f.keys() if k not in objects]
                kinds = {k: f[k].attrs.get("pandas_type") for k in old}
            bad = [k for k in old if kinds[k] not in ("frame", "series", "frame_table")]
            if bad:
                raise H5Error(f"{path}: cannot carry over the non-frame objects {bad}")
            merged = OrderedDict((k, read_hdf(path, k)) for k in old)
            merged.update(objects)
            objects = merged
    img = _Image()
    img._vl_fix = []
    built = []
    for key, fr in objects.items():
        if not isinstance(fr, Frame):
            fr = Frame(fr)
        names = list(fr.keys())
        cols = [np.asarray(fr[c]) for c in names]
        n = fr.n_rows
        if fr.series:
            if len(names) != 1 or cols[0].ndim != 1:
                raise ValueError(f"{key}: a series holds exactly one 1-D column")
            v = cols[0].astype(object) if cols[0].dtype.kind in "US" else cols[0]
            sname = getattr(fr, "series_name", names[0])
            children = {}
            gattrs = list(_GROUP_ATTRS) + [("pandas_type", "series"), ("pandas_version", "0.15.2"), ("encoding", "UTF-8"),
                                           ("errors", "strict")]
            _write_axis(img, children, gattrs, "index", fr.index, fr.index_names, n)
            gattrs.append(("name", sname if isinstance(sname, str) and sname else _Pickled(sname)))
            children["values"] = (_write_array(img, v), None)
            built.append((key.strip("/"), children, gattrs))
            continue
        for c, v in zip(names, cols):
            if v.ndim != 1 or len(v) != n:
                raise ValueError(f"{key}: column {c!r} has shape {v.shape}, expected ({n},)")
        cols = [v.astype(object) if v.dtype.kind in "US" else v for v in cols]
        children = {}
        gattrs = list(_GROUP_ATTRS) + [("pandas_type", "frame"), ("pandas_version", "0.15.2"), ("encoding", "UTF-8"),
                                       ("errors", "strict"), ("ndim", 2)]
        _write_index(img, children, "axis0", np.array(names, dtype=object), None)
        gattrs.append(("axis0_variety", "regular"))
        _write_axis(img, children, gattrs, "axis1", fr.index, fr.index_names, n)
        blocks = OrderedDict()
        for c, v in zip(names, cols):
            blocks.setdefault(v.dtype.str, []).append(c)
        gattrs.append(("nblocks", len(blocks)))
        for i, (_, items) in enumerate(blocks.items()):
            vals = np.stack([cols[names.index(c)] for c in items]) if n or True else None
            _write_index(img, children, f"block{i}_items", np.array(items, dtype=object), None)
            gattrs.append((f"block{i}_items_variety", "regular"))
            children[f"block{i}_values"] = (_write_array(img, vals), None)
        built.append((key.strip("/"), children, gattrs))
    leaf_k = max([4] + [(len(ch) + 1) // 2 for _, ch, _ in built] + [(len(built) + 1) // 2])
    top = {}
    for key, children, gattrs in built:
        if "/" in key:
            raise ValueError(f"nested key {key!r} is not supported")
        hdr, bt, heap = img.group(children, gattrs, leaf_k)
        top[key] = (hdr, (bt, heap))
    root = img.group(top, _GROUP_ATTRS + [("PYTABLES_FORMAT_VERSION", "2.1")], leaf_k)
    data = img.finish(root, leaf_k)
    with open(path, "wb") as fh:
        fh.write(data)
