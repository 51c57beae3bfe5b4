"""Test infrastructure: a small ctypes binding of the REAL HDF5 library that ships in this image
(/opt/conda/lib/libhdf5.so.103 = HDF5 1.10.6, with /opt/conda/bin/h5dump beside it), used to pin
building_detection_amd/h5lite.py against an independent implementation in both directions (SURVEY row
f-3).  h5py itself is not installed; the calls below are the ones h5py makes for Keras' save_weights / load_weights
(tf.keras saving/hdf5_format.py: save_weights_to_hdf5_group / load_weights_from_hdf5_group): fixed-length NULLPAD string
array attributes `layer_names` / `weight_names`, scalar string attributes `backend` / `keras_version`, one contiguous
little-endian float32 dataset per weight under /<layer>/<layer>/<weight>:0.

Never imported by the product.  `available()` is False on a box without the library (the GPU box has the same image, but
the tests skip rather than fail)."""
from __future__ import annotations

import ctypes as C
import os
import shutil
from typing import Dict, List, Tuple

import numpy as np

LIB_CANDIDATES = ("/opt/conda/lib/libhdf5.so.103", "/opt/conda/lib/libhdf5.so")
H5DUMP = next((p for p in ("/opt/conda/bin/h5dump", shutil.which("h5dump") or "") if p and os.path.exists(p)), None)

hid_t = C.c_int64
hsize_t = C.c_uint64
herr_t = C.c_int
H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5S_SCALAR = 0
H5T_INTEGER, H5T_FLOAT, H5T_STRING = 0, 1, 3
H5T_STR_NULLTERM, H5T_STR_NULLPAD = 0, 1
H5T_VARIABLE = C.c_size_t(-1).value
H5_INDEX_NAME, H5_ITER_INC = 0, 0
H5O_TYPE_GROUP, H5O_TYPE_DATASET = 0, 1

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = next((p for p in LIB_CANDIDATES if os.path.exists(p)), None)
        if path is None:
            raise OSError("no libhdf5 in this image")
        h = C.CDLL(path)
        h.H5open()
        sig = {
            "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]), "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]),
            "H5Fclose": (herr_t, [hid_t]),
            "H5Gcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]), "H5Gopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
            "H5Gclose": (herr_t, [hid_t]), "H5Gget_info": (herr_t, [hid_t, C.c_void_p]),
            "H5Lget_name_by_idx": (C.c_ssize_t, [hid_t, C.c_char_p, C.c_int, C.c_int, hsize_t, C.c_char_p, C.c_size_t, hid_t]),
            "H5Oopen": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Oclose": (herr_t, [hid_t]), "H5Iget_type": (C.c_int, [hid_t]),
            "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]), "H5Screate": (hid_t, [C.c_int]),
            "H5Sclose": (herr_t, [hid_t]), "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
            "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
            "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
            "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Dclose": (herr_t, [hid_t]),
            "H5Dwrite": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            "H5Dread": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            "H5Dget_space": (hid_t, [hid_t]), "H5Dget_type": (hid_t, [hid_t]), "H5Dget_create_plist": (hid_t, [hid_t]),
            "H5Pget_layout": (C.c_int, [hid_t]),
            "H5Acreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]), "H5Aopen": (hid_t, [hid_t, C.c_char_p, hid_t]),
            "H5Aclose": (herr_t, [hid_t]), "H5Awrite": (herr_t, [hid_t, hid_t, C.c_void_p]),
            "H5Aread": (herr_t, [hid_t, hid_t, C.c_void_p]), "H5Aexists": (C.c_int, [hid_t, C.c_char_p]),
            "H5Aget_type": (hid_t, [hid_t]), "H5Aget_space": (hid_t, [hid_t]), "H5Aget_num_attrs": (C.c_int, [hid_t]),
            "H5Aget_name": (C.c_ssize_t, [hid_t, C.c_size_t, C.c_char_p]), "H5Aopen_by_idx": (hid_t, [hid_t, C.c_char_p, C.c_int, C.c_int, hsize_t, hid_t, hid_t]),
            "H5Tcopy": (hid_t, [hid_t]), "H5Tclose": (herr_t, [hid_t]), "H5Tset_size": (herr_t, [hid_t, C.c_size_t]),
            "H5Tset_strpad": (herr_t, [hid_t, C.c_int]), "H5Tget_class": (C.c_int, [hid_t]), "H5Tget_size": (C.c_size_t, [hid_t]),
            "H5Tget_order": (C.c_int, [hid_t]), "H5Tis_variable_str": (C.c_int, [hid_t]), "H5Tset_cset": (herr_t, [hid_t, C.c_int]),
            "H5Pcreate": (hid_t, [hid_t]), "H5Pclose": (herr_t, [hid_t]), "H5Pset_create_intermediate_group": (herr_t, [hid_t, C.c_uint]),
            "H5Pset_libver_bounds": (herr_t, [hid_t, C.c_int, C.c_int]),
            "H5Dvlen_reclaim": (herr_t, [hid_t, hid_t, hid_t, C.c_void_p]),
            "H5Eset_auto2": (herr_t, [hid_t, C.c_void_p, C.c_void_p]),
        }
        for name, (res, args) in sig.items():
            f = getattr(h, name)
            f.restype, f.argtypes = res, args
        h.H5Eset_auto2(0, None, None)   # failures come back as negative ids, checked below; no stack dump on stderr
        _lib = h
    return _lib


def available() -> bool:
    try:
        lib()
        return True
    except OSError:
        return False


def version() -> Tuple[int, int, int]:
    a, b, c = C.c_uint(), C.c_uint(), C.c_uint()
    lib().H5get_libversion(C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def _g(name) -> int:
    """a predefined identifier (H5T_IEEE_F32LE, H5P_LINK_CREATE, ...): a global hid_t the library fills in H5open()"""
    return hid_t.in_dll(lib(), name).value


def _ok(v, what):
    if v < 0:
        raise OSError(f"libhdf5: {what} failed ({v})")
    return v


_NP2H5 = {np.dtype("<f4"): "H5T_IEEE_F32LE_g", np.dtype("<f8"): "H5T_IEEE_F64LE_g", np.dtype("<i4"): "H5T_STD_I32LE_g",
          np.dtype("<i8"): "H5T_STD_I64LE_g", np.dtype("u1"): "H5T_STD_U8LE_g"}


class Writer:
    """What h5py does for Keras, call for call; default file-creation properties = libver 'earliest' (the old-style
    symbol-table groups, version-1 object headers and B-trees every real Keras .h5 file has); `latest=True` asks for the
    newest format instead (h5py's libver='latest': superblock 3, link messages / dense groups)."""

    def __init__(self, path: str, latest: bool = False):
        h = lib()
        fapl = 0
        if latest:
            fapl = _ok(h.H5Pcreate(_g("H5P_CLS_FILE_ACCESS_ID_g")), "H5Pcreate(fapl)")
            _ok(h.H5Pset_libver_bounds(fapl, 2, 2), "H5Pset_libver_bounds")   # H5F_LIBVER_LATEST == H5F_LIBVER_V110 == 2 in 1.10.6
        self.f = _ok(h.H5Fcreate(path.encode(), H5F_ACC_TRUNC, 0, fapl), f"H5Fcreate({path})")
        if fapl:
            h.H5Pclose(fapl)
        self.lcpl = _ok(h.H5Pcreate(_g("H5P_CLS_LINK_CREATE_ID_g")), "H5Pcreate(lcpl)")
        _ok(h.H5Pset_create_intermediate_group(self.lcpl, 1), "H5Pset_create_intermediate_group")

    def group(self, path: str):
        h = lib()
        h.H5Gclose(_ok(h.H5Gcreate2(self.f, path.encode(), self.lcpl, 0, 0), f"H5Gcreate2({path})"))

    def dataset(self, path: str, a: np.ndarray):
        h = lib()
        a = np.asarray(a).copy(order="C")   # (np.ascontiguousarray would turn a 0-d array into shape (1,))
        dims = (hsize_t * max(a.ndim, 1))(*a.shape)
        sp = _ok(h.H5Screate_simple(a.ndim, dims, None) if a.ndim else h.H5Screate(H5S_SCALAR), "H5Screate")
        t = _g(_NP2H5[a.dtype.newbyteorder("<") if a.dtype.byteorder == ">" else a.dtype])
        d = _ok(h.H5Dcreate2(self.f, path.encode(), t, sp, self.lcpl, 0, 0), f"H5Dcreate2({path})")
        if a.size:
            _ok(h.H5Dwrite(d, t, 0, 0, 0, a.ctypes.data_as(C.c_void_p)), f"H5Dwrite({path})")
        h.H5Dclose(d)
        h.H5Sclose(sp)

    def _obj(self, path):
        return _ok(lib().H5Oopen(self.f, (path or "/").encode(), 0), f"H5Oopen({path})")

    def attr_strings(self, path: str, name: str, items: List[bytes]):
        """a 1-D array of fixed-length, null-padded strings: numpy 'S<n>' through h5py (layer_names / weight_names)"""
        h = lib()
        arr = np.array(items) if items else np.array([], dtype="S1")
        t = _ok(h.H5Tcopy(_g("H5T_C_S1_g")), "H5Tcopy")
        h.H5Tset_size(t, arr.dtype.itemsize)
        h.H5Tset_strpad(t, H5T_STR_NULLPAD)
        dims = (hsize_t * 1)(len(arr))
        sp = _ok(h.H5Screate_simple(1, dims, None), "H5Screate_simple")
        o = self._obj(path)
        a = _ok(h.H5Acreate2(o, name.encode(), t, sp, 0, 0), f"H5Acreate2({path}@{name}, {arr.nbytes} bytes)")
        if len(arr):
            _ok(h.H5Awrite(a, t, arr.ctypes.data_as(C.c_void_p)), "H5Awrite")
        h.H5Aclose(a), h.H5Oclose(o), h.H5Sclose(sp), h.H5Tclose(t)

    def attr_scalar_string(self, path: str, name: str, value: bytes, variable: bool = False):
        """bytes through h5py 2.x: a fixed-length NULLPAD scalar; str through h5py 3.x: a variable-length UTF-8 scalar"""
        h = lib()
        t = _ok(h.H5Tcopy(_g("H5T_C_S1_g")), "H5Tcopy")
        sp = _ok(h.H5Screate(H5S_SCALAR), "H5Screate")
        o = self._obj(path)
        if variable:
            h.H5Tset_size(t, H5T_VARIABLE)
            h.H5Tset_cset(t, 1)   # H5T_CSET_UTF8
            a = _ok(h.H5Acreate2(o, name.encode(), t, sp, 0, 0), "H5Acreate2")
            buf = C.c_char_p(value)
            _ok(h.H5Awrite(a, t, C.byref(buf)), "H5Awrite")
        else:
            h.H5Tset_size(t, max(len(value), 1))
            h.H5Tset_strpad(t, H5T_STR_NULLPAD)
            a = _ok(h.H5Acreate2(o, name.encode(), t, sp, 0, 0), "H5Acreate2")
            _ok(h.H5Awrite(a, t, C.c_char_p(value)), "H5Awrite")
        h.H5Aclose(a), h.H5Oclose(o), h.H5Sclose(sp), h.H5Tclose(t)

    def attr_array(self, path: str, name: str, arr: np.ndarray):
        h = lib()
        arr = np.asarray(arr).copy(order="C")
        t = _g(_NP2H5[arr.dtype])
        dims = (hsize_t * max(arr.ndim, 1))(*arr.shape)
        sp = _ok(h.H5Screate_simple(arr.ndim, dims, None) if arr.ndim else h.H5Screate(H5S_SCALAR), "H5Screate")
        o = self._obj(path)
        a = _ok(h.H5Acreate2(o, name.encode(), t, sp, 0, 0), "H5Acreate2")
        _ok(h.H5Awrite(a, t, arr.ctypes.data_as(C.c_void_p)), "H5Awrite")
        h.H5Aclose(a), h.H5Oclose(o), h.H5Sclose(sp)

    def close(self):
        h = lib()
        h.H5Pclose(self.lcpl)
        _ok(h.H5Fclose(self.f), "H5Fclose")


class _GInfo(C.Structure):
    _fields_ = [("storage_type", C.c_int), ("nlinks", hsize_t), ("max_corder", C.c_int64), ("mounted", C.c_int)]


class Reader:
    """What a real HDF5 library makes of a file: the links of a group, a dataset's class / size / byte order / shape /
    layout / values, an object's attributes."""

    def __init__(self, path: str):
        self.f = _ok(lib().H5Fopen(path.encode(), H5F_ACC_RDONLY, 0), f"H5Fopen({path})")

    def close(self):
        _ok(lib().H5Fclose(self.f), "H5Fclose")

    def keys(self, path: str = "/") -> List[str]:
        h = lib()
        g = _ok(h.H5Gopen2(self.f, (path or "/").encode(), 0), f"H5Gopen2({path})")
        info = _GInfo()
        _ok(h.H5Gget_info(g, C.byref(info)), "H5Gget_info")
        out = []
        for i in range(info.nlinks):
            n = _ok(h.H5Lget_name_by_idx(g, b".", H5_INDEX_NAME, H5_ITER_INC, i, None, 0, 0), "H5Lget_name_by_idx")
            buf = C.create_string_buffer(n + 1)
            h.H5Lget_name_by_idx(g, b".", H5_INDEX_NAME, H5_ITER_INC, i, buf, n + 1, 0)
            out.append(buf.value.decode())
        h.H5Gclose(g)
        return out

    def is_dataset(self, path: str) -> bool:
        h = lib()
        o = _ok(h.H5Oopen(self.f, path.encode(), 0), f"H5Oopen({path})")
        t = h.H5Iget_type(o)   # H5I_GROUP = 2, H5I_DATASET = 5
        h.H5Oclose(o)
        return t == 5

    def _space_dims(self, sp) -> Tuple[int, ...]:
        h = lib()
        nd = h.H5Sget_simple_extent_ndims(sp)
        dims = (hsize_t * max(nd, 1))()
        if nd > 0:
            h.H5Sget_simple_extent_dims(sp, dims, None)
        return tuple(int(dims[i]) for i in range(nd))

    def dataset(self, path: str) -> Dict[str, object]:
        """-> {'class', 'size', 'little_endian', 'shape', 'contiguous', 'data'} (data read as native float / int of that size)"""
        h = lib()
        d = _ok(h.H5Dopen2(self.f, path.encode(), 0), f"H5Dopen2({path})")
        t, sp, pl = h.H5Dget_type(d), h.H5Dget_space(d), h.H5Dget_create_plist(d)
        cls, size, order = h.H5Tget_class(t), h.H5Tget_size(t), h.H5Tget_order(t)
        shape = self._space_dims(sp)
        npdt = {(H5T_FLOAT, 4): "<f4", (H5T_FLOAT, 8): "<f8", (H5T_FLOAT, 2): "<f2", (H5T_INTEGER, 4): "<i4", (H5T_INTEGER, 8): "<i8",
                (H5T_INTEGER, 1): "u1"}[(cls, size)]
        a = np.empty(shape, npdt)
        if a.size:
            memt = {"<f4": "H5T_IEEE_F32LE_g", "<f8": "H5T_IEEE_F64LE_g", "<i4": "H5T_STD_I32LE_g", "<i8": "H5T_STD_I64LE_g",
                    "u1": "H5T_STD_U8LE_g", "<f2": None}[npdt]
            _ok(h.H5Dread(d, t if memt is None else _g(memt), 0, 0, 0, a.ctypes.data_as(C.c_void_p)), f"H5Dread({path})")
        out = {"class": cls, "size": size, "little_endian": order == 0, "shape": shape, "contiguous": h.H5Pget_layout(pl) == 1, "data": a}
        h.H5Pclose(pl), h.H5Sclose(sp), h.H5Tclose(t), h.H5Dclose(d)
        return out

    def attr_names(self, path: str = "/") -> List[str]:
        h = lib()
        o = _ok(h.H5Oopen(self.f, (path or "/").encode(), 0), f"H5Oopen({path})")
        out = []
        for i in range(_ok(h.H5Aget_num_attrs(o), "H5Aget_num_attrs")):
            a = _ok(h.H5Aopen_by_idx(o, b".", H5_INDEX_NAME, H5_ITER_INC, i, 0, 0), "H5Aopen_by_idx")
            n = h.H5Aget_name(a, 0, None)
            buf = C.create_string_buffer(n + 1)
            h.H5Aget_name(a, n + 1, buf)
            out.append(buf.value.decode())
            h.H5Aclose(a)
        h.H5Oclose(o)
        return out

    def attr(self, path: str, name: str):
        """strings -> bytes / list of bytes (trailing NULs stripped, as h5py / numpy 'S' do); numbers -> numpy array"""
        h = lib()
        o = _ok(h.H5Oopen(self.f, (path or "/").encode(), 0), f"H5Oopen({path})")
        a = _ok(h.H5Aopen(o, name.encode(), 0), f"H5Aopen({path}@{name})")
        t, sp = h.H5Aget_type(a), h.H5Aget_space(a)
        shape = self._space_dims(sp)
        n = int(np.prod(shape)) if shape else 1
        cls, size = h.H5Tget_class(t), h.H5Tget_size(t)
        if cls == H5T_STRING:
            if h.H5Tis_variable_str(t) > 0:
                ptrs = (C.c_char_p * n)()
                _ok(h.H5Aread(a, t, ptrs), "H5Aread")
                vals = [bytes(p) if p is not None else b"" for p in ptrs]
                h.H5Dvlen_reclaim(t, sp, 0, ptrs)
            else:
                buf = C.create_string_buffer(max(n * size, 1))
                if n:
                    _ok(h.H5Aread(a, t, buf), "H5Aread")
                vals = [buf.raw[i * size:(i + 1) * size].rstrip(b"\0") for i in range(n)]
            out = vals if shape else vals[0]
        else:
            npdt = {(H5T_FLOAT, 4): "<f4", (H5T_FLOAT, 8): "<f8", (H5T_INTEGER, 4): "<i4", (H5T_INTEGER, 8): "<i8", (H5T_INTEGER, 1): "u1"}[(cls, size)]
            out = np.empty(shape, npdt)
            _ok(h.H5Aread(a, t, out.ctypes.data_as(C.c_void_p)), "H5Aread")
        h.H5Sclose(sp), h.H5Tclose(t), h.H5Aclose(a), h.H5Oclose(o)
        return out
