"""A minimal HDF5 reader / writer for Keras weight files (h5py is not available in this image, libhdf5 only as a test-side checker).

The reference loads `resnet34.h5 ... bam.h5` with `model.load_weights` (predict.py:21-49) and writes
`epoch_N_weights.h5` with `model.save_weights` (train_model/DeepLabv3plus.py:778-780).  Those are HDF5 files in the
layout h5py writes by default (libver "earliest"), which is all this module covers:

    read    superblock v0 (and v2 / v3), object headers v1 (and v2), old-style groups (symbol table: v1 B-tree + local
            heap + SNOD nodes) and new-style compact groups (link messages), attributes (message 0x0C, versions 1-3),
            dataspaces v1 / v2, datatypes: fixed point, IEEE float, fixed-length string, variable-length string (global
            heap); data layouts: compact and contiguous (v3 / v4).  Chunked / filtered datasets raise NotImplementedError
            (Keras writes its weights contiguous and uncompressed).
    write   superblock v0, object headers v1, old-style groups, contiguous little-endian datasets, attributes v1 holding
            scalars, numeric arrays or arrays of fixed-length byte strings - the subset `keras save_weights` produces.

Written from the HDF5 File Format Specification (version 3.0 of the specification, sections II-IV).  PINNING: h5py and Keras
are not in this image, but a real HDF5 library is (/opt/conda/lib/libhdf5.so.103 = HDF5 1.10.6, with h5dump) - round 4's
note here said otherwise and was wrong.  tests/test_h5_libhdf5_cpu.py binds it with ctypes (tests/_libhdf5.py) and checks
both directions at the scale of the real models: files this writer produces for all five graphs (DeepLabv3+: 203 layer
groups, 652 datasets) are opened by libhdf5 and every dataset (class, size, byte order, contiguous layout, shape, values)
and name-list attribute compared, and h5dump walks them; Keras-2-layout files assembled by libhdf5 with the calls h5py
makes (old format with multi-node symbol-table B-trees, name lists split over layer_names0.., the `model.save()` layout,
h5py-3 variable-length strings, the newest file format with compact groups) are loaded by this reader; densely stored
groups (newest format, > 8 links: never what Keras' default writes) are refused with the reason.  Beside that: write ->
read round trips and two hand-assembled byte-level fixtures (tests/test_weights_io_cpu.py).  What remains unverified is a
file written by Keras itself (none exists here).
"""
from __future__ import annotations

import struct
from typing import Dict, List, Optional, Tuple, Union

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(OSError):
    """Raised for anything that is not a readable HDF5 file of the supported subset (an OSError, like h5py's)."""


# ===================================================================================================== reading
class Dataset:
    def __init__(self, f: "File", shape, dtype, layout, name):
        self._f, self.shape, self.dtype, self._layout, self.name = f, tuple(shape), dtype, layout, name
        self.attrs: Dict[str, object] = {}

    def __array__(self, dtype=None, copy=None):
        a = self[()]
        return a.astype(dtype) if dtype is not None else a

    def __getitem__(self, key):
        kind = self._layout[0]
        n = int(np.prod(self.shape)) if self.shape else 1
        if kind == "compact":
            raw = self._layout[1]
        elif kind == "contiguous":
            addr, size = self._layout[1], self._layout[2]
            raw = b"" if addr == UNDEF else self._f._read(addr, n * self.dtype.itemsize if size is None else size)
        else:
            raise NotImplementedError(f"{self.name}: {kind} dataset layout (chunked / filtered data) is outside the subset "
                                      "this reader covers - Keras writes weights contiguous and uncompressed")
        if len(raw) < n * self.dtype.itemsize:  # never written: HDF5's default fill value is zero
            raw = raw + b"\0" * (n * self.dtype.itemsize - len(raw))
        a = np.frombuffer(raw, dtype=self.dtype, count=n).reshape(self.shape).copy()
        return a if key == () or key is Ellipsis else a[key]


class Group:
    def __init__(self, f: "File", name: str):
        self._f, self.name = f, name
        self.attrs: Dict[str, object] = {}
        self._links: Dict[str, int] = {}
        self._cache: Dict[str, object] = {}

    def keys(self):
        return list(self._links)

    def __contains__(self, k):
        try:
            self[k]
            return True
        except KeyError:
            return False

    def __iter__(self):
        return iter(self._links)

    def __getitem__(self, path: str):
        node = self
        for part in [p for p in path.split("/") if p]:
            if not isinstance(node, Group) or part not in node._links:
                raise KeyError(f"{path!r}: no object {part!r} in group {getattr(node, 'name', '?')!r}")
            if part not in node._cache:
                child = (node.name.rstrip("/") + "/" + part)
                node._cache[part] = node._f._object(node._links[part], child)
            node = node._cache[part]
        return node


class File(Group):
    """`h5lite.File(path)` - read-only view with the h5py spelling the weight loader needs: `f.attrs[...]`, `f[name]`,
    `group.attrs`, `np.asarray(dataset)`."""

    def __init__(self, path: str):
        with open(path, "rb") as fh:
            self._buf = fh.read()
        Group.__init__(self, self, "/")
        self._base = 0
        self._so = self._sl = 8
        root = self._superblock()
        obj = self._object(root, "/")
        if not isinstance(obj, Group):
            raise H5Error(f"{path}: the root object is not a group")
        self.attrs, self._links = obj.attrs, obj._links

    def close(self):
        self._buf = b""

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- primitives
    def _read(self, addr: int, n: int) -> bytes:
        a = self._base + addr
        if addr == UNDEF or a < 0 or a + n > len(self._buf):
            raise H5Error(f"HDF5: read of {n} bytes at {addr:#x} beyond the end of the file ({len(self._buf)} bytes) - truncated file?")
        return self._buf[a:a + n]

    def _u(self, addr: int, n: int) -> int:
        return int.from_bytes(self._read(addr, n), "little")

    def _superblock(self) -> int:
        pos = -1
        off = 0
        while off < len(self._buf):  # the signature sits at 0, 512, 1024, 2048, ...
            if self._buf[off:off + 8] == SIGNATURE:
                pos = off
                break
            off = 512 if off == 0 else off * 2
        if pos < 0:
            raise H5Error("not an HDF5 file (signature \\x89HDF\\r\\n\\x1a\\n not found)")
        ver = self._buf[pos + 8]
        if ver in (0, 1):
            self._so, self._sl = self._buf[pos + 13], self._buf[pos + 14]
            p = pos + 24 + (4 if ver == 1 else 0)
            self._base = int.from_bytes(self._buf[p:p + self._so], "little")
            if self._base == UNDEF & ((1 << (8 * self._so)) - 1):
                self._base = 0
            p += 4 * self._so  # base, free-space, end-of-file, driver-info addresses
            # root group symbol table entry: link name offset, object header address, cache type, reserved, scratch
            return int.from_bytes(self._buf[p + self._so:p + 2 * self._so], "little")
        if ver in (2, 3):
            self._so, self._sl = self._buf[pos + 9], self._buf[pos + 10]
            p = pos + 12
            self._base = int.from_bytes(self._buf[p:p + self._so], "little")
            p += 3 * self._so  # base, superblock extension, end-of-file
            return int.from_bytes(self._buf[p:p + self._so], "little")
        raise H5Error(f"HDF5 superblock version {ver} is not supported")

    # ---- object headers
    def _messages(self, addr: int) -> List[Tuple[int, bytes]]:
        out: List[Tuple[int, bytes]] = []
        if self._read(addr, 4) == b"OHDR":
            flags = self._u(addr + 5, 1)
            p = addr + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            nsz = 1 << (flags & 3)
            size0 = self._u(p, nsz)
            p += nsz
            chunks = [(p, size0)]
            order = bool(flags & 0x04)
            while chunks:
                cp, csz = chunks.pop(0)
                end = cp + csz
                while cp + 4 <= end:
                    mtype, msize, _mflags = self._u(cp, 1), self._u(cp + 1, 2), self._u(cp + 3, 1)
                    cp += 4 + (2 if order else 0)
                    body = self._read(cp, msize)
                    cp += msize
                    if mtype == 0x10:
                        caddr = int.from_bytes(body[:self._so], "little")
                        clen = int.from_bytes(body[self._so:self._so + self._sl], "little")
                        if self._read(caddr, 4) != b"OCHK":
                            raise H5Error("HDF5: object header continuation without OCHK signature")
                        chunks.append((caddr + 4, clen - 8))  # signature in front, checksum behind
                    elif mtype != 0:
                        out.append((mtype, body))
            return out
        ver = self._u(addr, 1)
        if ver != 1:
            raise H5Error(f"HDF5: object header version {ver} at {addr:#x} is not supported")
        nmsg, hsize = self._u(addr + 2, 2), self._u(addr + 8, 4)
        chunks = [(addr + 16, hsize)]
        while chunks and len(out) < 100000:
            cp, csz = chunks.pop(0)
            end = cp + csz
            while cp + 8 <= end and nmsg > 0:
                mtype, msize = self._u(cp, 2), self._u(cp + 2, 2)
                body = self._read(cp + 8, msize)
                cp += 8 + msize
                nmsg -= 1
                if mtype == 0x10:
                    chunks.append((int.from_bytes(body[:self._so], "little"),
                                   int.from_bytes(body[self._so:self._so + self._sl], "little")))
                elif mtype != 0:
                    out.append((mtype, body))
        return out

    def _object(self, addr: int, name: str):
        msgs = self._messages(addr)
        types = {t for t, _ in msgs}
        attrs = {}
        for t, b in msgs:
            if t == 0x0C:
                k, v = self._attribute(b)
                attrs[k] = v
        for t, b in msgs:
            if t == 0x15:  # attribute info: version, flags, [max creation index], fractal heap address, ...
                q = 2 + (2 if b[1] & 1 else 0)
                if int.from_bytes(b[q:q + self._so], "little") != UNDEF & ((1 << (8 * self._so)) - 1):
                    raise NotImplementedError(f"{name}: densely stored attributes (fractal heap) are outside the subset this reader covers")
        if 0x08 in types:  # dataset
            shape: Tuple[int, ...] = ()
            dtype = None
            layout = None
            for t, b in msgs:
                if t == 0x01:
                    shape = self._dataspace(b)
                elif t == 0x03:
                    dtype, _ = self._datatype(b)
                elif t == 0x08:
                    layout = self._layout(b)
            if dtype is None or layout is None or isinstance(dtype, str):
                raise H5Error(f"{name}: dataset with an unsupported datatype or layout")
            ds = Dataset(self, shape, dtype, layout, name)
            ds.attrs = attrs
            return ds
        g = Group(self, name)
        g.attrs = attrs
        for t, b in msgs:
            if t == 0x11:  # symbol table: B-tree address, local heap address
                btree = int.from_bytes(b[:self._so], "little")
                heap = int.from_bytes(b[self._so:2 * self._so], "little")
                g._links.update(self._symbol_table(btree, heap))
            elif t == 0x06:
                k, a = self._link(b)
                if a is not None:
                    g._links[k] = a
            elif t == 0x02:
                p = 2 + (8 if b[1] & 1 else 0)
                fheap = int.from_bytes(b[p:p + self._so], "little")
                if fheap != UNDEF:
                    raise NotImplementedError(f"{name}: densely stored links (fractal heap) are outside the subset this reader covers")
        return g

    # ---- groups
    def _heap_string(self, heap_data: int, off: int) -> str:
        end = self._buf.index(b"\0", self._base + heap_data + off)
        return self._buf[self._base + heap_data + off:end].decode("utf-8")

    def _symbol_table(self, btree: int, heap: int) -> Dict[str, int]:
        if self._read(heap, 4) != b"HEAP":
            raise H5Error("HDF5: local heap without HEAP signature")
        data = self._u(heap + 8 + 2 * self._sl, self._so)
        out: Dict[str, int] = {}

        def walk(node):
            sig = self._read(node, 4)
            if sig == b"TREE":
                level, used = self._u(node + 5, 1), self._u(node + 6, 2)
                p = node + 8 + 2 * self._so
                for i in range(used):
                    child = self._u(p + self._sl + i * (self._sl + self._so), self._so)
                    walk(child)
                _ = level
            elif sig == b"SNOD":
                n = self._u(node + 6, 2)
                p = node + 8
                esz = 2 * self._so + 8 + 16
                for i in range(n):
                    noff = self._u(p + i * esz, self._so)
                    oaddr = self._u(p + i * esz + self._so, self._so)
                    out[self._heap_string(data, noff)] = oaddr
            else:
                raise H5Error(f"HDF5: unexpected node signature {sig!r} in a group B-tree")

        if btree != UNDEF:
            walk(btree)
        return out

    def _link(self, b: bytes):
        flags = b[1]
        p = 2
        ltype = 0
        if flags & 0x08:
            ltype = b[p]
            p += 1
        if flags & 0x04:
            p += 8
        if flags & 0x10:
            p += 1
        lsz = 1 << (flags & 3)
        nlen = int.from_bytes(b[p:p + lsz], "little")
        p += lsz
        name = b[p:p + nlen].decode("utf-8")
        p += nlen
        if ltype != 0:
            return name, None  # soft / external links: not followed
        return name, int.from_bytes(b[p:p + self._so], "little")

    # ---- messages
    def _dataspace(self, b: bytes) -> Tuple[int, ...]:
        ver, rank, flags = b[0], b[1], b[2]
        p = 8 if ver == 1 else 4
        if ver == 2 and b[3] == 2:
            return (0,)  # null dataspace
        return tuple(int.from_bytes(b[p + i * self._sl:p + (i + 1) * self._sl], "little") for i in range(rank))

    def _datatype(self, b: bytes):
        """-> (numpy dtype | "vlen_str" | ("vlen", base dtype), bytes consumed)"""
        cls, ver = b[0] & 0x0F, b[0] >> 4
        bits = b[1] | (b[2] << 8) | (b[3] << 16)
        size = int.from_bytes(b[4:8], "little")
        order = ">" if bits & 1 else "<"
        if cls == 0:
            signed = bool(bits & 0x08)
            return np.dtype(f"{order}{'i' if signed else 'u'}{size}"), 8 + 4
        if cls == 1:
            if size not in (2, 4, 8):
                raise H5Error(f"HDF5: {size}-byte floating point type")
            return np.dtype(f"{order}f{size}"), 8 + 12
        if cls == 3:
            return np.dtype(f"S{size}"), 8
        if cls == 9:
            base, used = self._datatype(b[8:])
            if (bits & 0x0F) == 1:
                return "vlen_str", 8 + used
            return ("vlen", base), 8 + used
        if cls == 4:  # bitfield: as unsigned
            return np.dtype(f"{order}u{size}"), 8 + 4
        raise H5Error(f"HDF5: datatype class {cls} (version {ver}) is not supported")

    def _layout(self, b: bytes):
        ver = b[0]
        if ver in (3, 4):
            cls = b[1]
            if cls == 0:
                n = int.from_bytes(b[2:4], "little")
                return ("compact", bytes(b[4:4 + n]))
            if cls == 1:
                return ("contiguous", int.from_bytes(b[2:2 + self._so], "little"),
                        int.from_bytes(b[2 + self._so:2 + self._so + self._sl], "little"))
            return ("chunked",)
        if ver in (1, 2):
            rank, cls = b[1], b[2]
            if cls == 1:
                return ("contiguous", int.from_bytes(b[8:8 + self._so], "little"), None)
            if cls == 0:
                p = 8 + 4 * rank
                n = int.from_bytes(b[p:p + 4], "little")
                return ("compact", bytes(b[p + 4:p + 4 + n]))
            return ("chunked",)
        raise H5Error(f"HDF5: data layout message version {ver}")

    def _global_heap_object(self, addr: int, index: int) -> bytes:
        if self._read(addr, 4) != b"GCOL":
            raise H5Error("HDF5: global heap collection without GCOL signature")
        size = self._u(addr + 8, self._sl)
        p, end = addr + 8 + self._sl, addr + size
        while p + 8 + self._sl <= end:
            idx, osz = self._u(p, 2), self._u(p + 8, self._sl)
            if idx == 0:
                break
            if idx == index:
                return self._read(p + 8 + self._sl, osz)
            p += 8 + self._sl + (osz + 7) // 8 * 8
        raise H5Error(f"HDF5: object {index} not found in the global heap collection at {addr:#x}")

    def _attribute(self, b: bytes):
        ver = b[0]
        nsz, tsz, ssz = (int.from_bytes(b[2:4], "little"), int.from_bytes(b[4:6], "little"), int.from_bytes(b[6:8], "little"))
        p = 8 + (1 if ver == 3 else 0)
        pad = (lambda n: (n + 7) // 8 * 8) if ver == 1 else (lambda n: n)
        name = b[p:p + nsz].split(b"\0")[0].decode("utf-8")
        p += pad(nsz)
        dt, _ = self._datatype(b[p:p + tsz])
        p += pad(tsz)
        shape = self._dataspace(b[p:p + ssz])
        p += pad(ssz)
        n = int(np.prod(shape)) if shape else 1
        if dt == "vlen_str":
            vals = []
            for i in range(n):
                q = p + i * (4 + self._so + 4)
                ln = int.from_bytes(b[q:q + 4], "little")
                ga = int.from_bytes(b[q + 4:q + 4 + self._so], "little")
                gi = int.from_bytes(b[q + 4 + self._so:q + 8 + self._so], "little")
                vals.append(self._global_heap_object(ga, gi)[:ln] if ln else b"")
            arr = np.array(vals, dtype=object)
            return name, (arr.reshape(shape) if shape else arr[0])
        if isinstance(dt, tuple):
            raise NotImplementedError(f"attribute {name!r}: variable-length sequences are outside the subset this reader covers")
        arr = np.frombuffer(b[p:p + n * dt.itemsize], dtype=dt, count=n).copy()
        return name, (arr.reshape(shape) if shape else arr[0])


# ===================================================================================================== writing
def _dtype_message(dt: np.dtype) -> bytes:
    dt = np.dtype(dt)
    if dt.kind == "f":
        # IEEE little-endian: class 1 version 1; bits: byte order 0, padding 0, mantissa normalisation 2 (implied msb),
        # sign location in bits 8-15
        spec = {2: (15, 10, 5, 0, 10, 15), 4: (31, 23, 8, 0, 23, 127), 8: (63, 52, 11, 0, 52, 1023)}[dt.itemsize]
        sign, eloc, esz, mloc, msz, bias = spec
        bits = 0x20 | (sign << 8)
        return (bytes([0x11, bits & 0xFF, (bits >> 8) & 0xFF, (bits >> 16) & 0xFF]) + struct.pack("<I", dt.itemsize) +
                struct.pack("<HHBBBBI", 0, dt.itemsize * 8, eloc, esz, mloc, msz, bias))
    if dt.kind in "iu":
        bits = 0x08 if dt.kind == "i" else 0
        return bytes([0x10, bits, 0, 0]) + struct.pack("<I", dt.itemsize) + struct.pack("<HH", 0, dt.itemsize * 8)
    if dt.kind == "S":
        # fixed-length string, null padded (type 1), ASCII character set 0 (what h5py writes for numpy 'S' data)
        return bytes([0x13, 0x01, 0, 0]) + struct.pack("<I", max(dt.itemsize, 1))
    raise TypeError(f"h5lite: dtype {dt} cannot be written")


def _dataspace_message(shape: Tuple[int, ...]) -> bytes:
    # version 1: version, rank, flags (bit 0: max dims present), reserved[5], dims, max dims (h5py writes them: = dims)
    shape = tuple(int(s) for s in shape)
    out = bytes([1, len(shape), 1 if shape else 0]) + b"\0" * 5
    for s in shape:
        out += struct.pack("<Q", s)
    for s in shape:
        out += struct.pack("<Q", s)
    return out


def _pad8(b: bytes) -> bytes:
    return b + b"\0" * (-len(b) % 8)


def _attr_message(name: str, value) -> bytes:
    if isinstance(value, (bytes, str)):
        value = np.array(value.encode("utf-8") if isinstance(value, str) else value)  # 0-d 'S' array
    arr = np.asarray(value)
    if arr.dtype.kind == "U":
        arr = np.char.encode(arr, "utf-8")
    if arr.dtype.kind == "O":
        arr = np.array([v if isinstance(v, bytes) else str(v).encode("utf-8") for v in arr.reshape(-1)]).reshape(arr.shape)
    if arr.dtype.kind == "S" and arr.dtype.itemsize == 0:
        arr = arr.astype("S1")
    if arr.dtype.byteorder == ">":
        arr = arr.astype(arr.dtype.newbyteorder("<"))
    nm = name.encode("utf-8") + b"\0"
    dtm, dsm = _dtype_message(arr.dtype), _dataspace_message(arr.shape)
    body = bytes([1, 0]) + struct.pack("<HHH", len(nm), len(dtm), len(dsm)) + _pad8(nm) + _pad8(dtm) + _pad8(dsm) + arr.tobytes()
    if len(body) > 64000:
        raise ValueError(f"attribute {name!r}: {len(body)} bytes do not fit an object-header message (Keras splits such lists "
                         "into name0, name1, ...: weights_io does the same)")
    return body


class _WNode:
    def __init__(self):
        self.attrs: Dict[str, object] = {}


class _WGroup(_WNode):
    def __init__(self):
        super().__init__()
        self.children: Dict[str, _WNode] = {}


class _WDataset(_WNode):
    def __init__(self, arr: np.ndarray):
        super().__init__()
        arr = np.asarray(arr, order="C")  # (ascontiguousarray would turn a scalar into a 1-element vector)
        if arr.dtype.byteorder == ">":
            arr = arr.astype(arr.dtype.newbyteorder("<"))
        self.arr = arr


class Writer:
    """`w = Writer(); g = w.group("conv2d"); w.dataset("conv2d/conv2d/kernel:0", array); w.attrs("", "backend", b"...")`;
    `w.save(path)` lays the file out: superblock v0, every group an old-style symbol table (one B-tree node over SNOD
    leaves of LEAF_K * 2 entries), object headers v1, contiguous data."""

    INTERNAL_K = 16   # children per B-tree node <= 2 * INTERNAL_K (HDF5's default); every group is ONE such node here
    LEAF_K = 4        # symbols per SNOD <= 2 * LEAF_K (HDF5's default 4); raised by save() when a group has > 256 links:
                      # both are file-level parameters stored in the superblock

    def __init__(self):
        self.root = _WGroup()

    def _walk(self, path: str, create=True) -> _WNode:
        node: _WNode = self.root
        for part in [p for p in path.split("/") if p]:
            assert isinstance(node, _WGroup), f"{path}: {part} lies below a dataset"
            if part not in node.children:
                if not create:
                    raise KeyError(path)
                node.children[part] = _WGroup()
            node = node.children[part]
        return node

    def group(self, path: str) -> None:
        self._walk(path)

    def dataset(self, path: str, arr) -> None:
        parts = [p for p in path.split("/") if p]
        parent = self._walk("/".join(parts[:-1]))
        assert isinstance(parent, _WGroup)
        parent.children[parts[-1]] = _WDataset(np.asarray(arr))

    def attr(self, path: str, name: str, value) -> None:
        self._walk(path, create=False).attrs[name] = value

    # ---- layout
    def save(self, path: str) -> None:
        buf = bytearray()

        def widest(g: _WGroup) -> int:
            return max([len(g.children)] + [widest(c) for c in g.children.values() if isinstance(c, _WGroup)])

        self.LEAF_K = max(4, -(-widest(self.root) // (2 * 2 * self.INTERNAL_K)))

        def alloc(n: int, align: int = 8) -> int:
            buf.extend(b"\0" * (-len(buf) % align))
            a = len(buf)
            buf.extend(b"\0" * n)
            return a

        def put(a: int, data: bytes):
            buf[a:a + len(data)] = data

        sb = alloc(96)  # superblock v0: 56 bytes + root symbol table entry (40)

        def header(messages: List[Tuple[int, bytes]]) -> int:
            body = b""
            for t, m in messages:
                m = _pad8(m)
                body += struct.pack("<HHBBBB", t, len(m), 0, 0, 0, 0) + m
            a = alloc(16 + len(body))
            put(a, bytes([1, 0]) + struct.pack("<HII", len(messages), 1, len(body)) + b"\0" * 4 + body)
            return a

        def write_node(node: _WNode) -> int:
            attrs = [(0x0C, _attr_message(k, v)) for k, v in node.attrs.items()]
            if isinstance(node, _WDataset):
                arr = node.arr
                daddr = alloc(max(arr.nbytes, 1)) if arr.nbytes else UNDEF
                if arr.nbytes:
                    put(daddr, arr.tobytes())
                layout = bytes([3, 1]) + struct.pack("<QQ", daddr, arr.nbytes)
                # fill value message v2: allocate late (2), write time "if set" (2), undefined fill
                fill = bytes([2, 2, 2, 0])
                return header([(0x01, _dataspace_message(arr.shape)), (0x03, _dtype_message(arr.dtype)), (0x05, fill),
                               (0x08, layout)] + attrs)
            assert isinstance(node, _WGroup)
            names = sorted(node.children)  # SNOD entries are ordered by name (byte-wise, as strcmp)
            child_addr = {n: write_node(node.children[n]) for n in names}
            # local heap: offset 0 holds the empty string; every name null-terminated, 8-byte aligned; one free block at the end
            heap_data = bytearray(b"\0" * 8)
            name_off = {}
            for n in names:
                name_off[n] = len(heap_data)
                heap_data += _pad8(n.encode("utf-8") + b"\0")
            free_off = len(heap_data)
            heap_data += struct.pack("<QQ", 1, 32) + b"\0" * 16   # free block: next = 1 (none), size 32
            hd = alloc(len(heap_data))
            put(hd, bytes(heap_data))
            heap = alloc(32)
            put(heap, b"HEAP" + bytes([0, 0, 0, 0]) + struct.pack("<QQQ", len(heap_data), free_off, hd))
            # SNOD leaves
            per = 2 * self.LEAF_K
            leaves = []
            for i in range(0, max(len(names), 1), per):
                chunk = names[i:i + per]
                sn = alloc(8 + per * 40)
                ent = b""
                for n in chunk:
                    ent += struct.pack("<QQII", name_off[n], child_addr[n], 0, 0) + b"\0" * 16
                put(sn, b"SNOD" + bytes([1, 0]) + struct.pack("<H", len(chunk)) + ent)
                leaves.append((sn, name_off[chunk[-1]] if chunk else 0))
            assert len(leaves) <= 2 * self.INTERNAL_K, "group too large for a single B-tree node"
            tree = alloc(24 + (2 * self.INTERNAL_K + 1) * 8 + 2 * self.INTERNAL_K * 8)
            body = b"TREE" + bytes([0, 0]) + struct.pack("<H", len(leaves)) + struct.pack("<QQ", UNDEF, UNDEF)
            body += struct.pack("<Q", 0)  # key 0: the empty string, smaller than every name
            for sn, last_off in leaves:
                body += struct.pack("<QQ", sn, last_off)  # child, then the key = largest name in that child
            put(tree, body)
            return header([(0x11, struct.pack("<QQ", tree, heap))] + attrs)

        root_addr = write_node(self.root)
        # cached scratch of the root entry: its B-tree and heap addresses (cache type 1)
        msgs = File.__new__(File)
        msgs._buf, msgs._base, msgs._so, msgs._sl = bytes(buf), 0, 8, 8
        stab = [b for t, b in msgs._messages(root_addr) if t == 0x11][0]
        eof = len(buf)
        head = (SIGNATURE + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + struct.pack("<HH", self.LEAF_K, self.INTERNAL_K) + struct.pack("<I", 0) +
                struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF) + struct.pack("<QQII", 0, root_addr, 1, 0) + stab[:16])
        put(sb, head)
        with open(path, "wb") as fh:
            fh.write(bytes(buf))


def is_hdf5(path: str) -> bool:
    try:
        with open(path, "rb") as fh:
            return fh.read(8) == SIGNATURE
    except OSError:
        return False
