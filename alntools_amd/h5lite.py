# -*- coding: utf-8 -*-
"""A small ctypes binding of libhdf5 (C API, 1.10.x) -- just what the EMASE ``.h5`` layout needs.

PyTables and h5py are not installed in this image; ``libhdf5.so`` is (``/opt/conda/lib``).  Files are written the way
PyTables writes them for the reference (``Sparse3DMatrix.py:325-342``, ``AlignmentPropertyMatrix.py:507-532``):
chunked datasets with shuffle + deflate level 1 (``tables.Filters(complevel=1, complib='zlib')``), numeric scalars as
native attributes, Python tuples/lists as pickled string attributes (what PyTables does for non-numpy objects), byte
strings as fixed-length ASCII attributes.  Whether PyTables reads these files exactly as its own is NOT verified here
(".h5 unpinned"); this module's own reader round-trips them and ``h5dump`` accepts them.
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import os
import pickle

import numpy as np

hid_t = C.c_int64
hsize_t = C.c_uint64
_lib = None


def _find():
    for cand in (os.environ.get("ALNTOOLS_LIBHDF5"), "/opt/conda/lib/libhdf5.so", ctypes.util.find_library("hdf5")):
        if cand and (os.path.exists(cand) or not os.path.isabs(cand)):
            try:
                return C.CDLL(cand)
            except OSError:
                continue
    raise RuntimeError("libhdf5 not found (set ALNTOOLS_LIBHDF5): EMASE .h5 I/O is unavailable; .bin output is complete")


def lib():
    global _lib
    if _lib is None:
        l = _find()
        l.H5open()
        for name, res, args in [
            ("H5Fcreate", hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]), ("H5Fopen", hid_t, [C.c_char_p, C.c_uint, hid_t]),
            ("H5Fclose", C.c_int, [hid_t]), ("H5Gcreate2", hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]),
            ("H5Gclose", C.c_int, [hid_t]), ("H5Screate_simple", hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
            ("H5Screate", hid_t, [C.c_int]), ("H5Sclose", C.c_int, [hid_t]), ("H5Pcreate", hid_t, [hid_t]),
            ("H5Pset_chunk", C.c_int, [hid_t, C.c_int, C.POINTER(hsize_t)]), ("H5Pset_deflate", C.c_int, [hid_t, C.c_uint]),
            ("H5Pset_shuffle", C.c_int, [hid_t]), ("H5Pclose", C.c_int, [hid_t]),
            ("H5Dcreate2", hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
            ("H5Dwrite", C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]), ("H5Dclose", C.c_int, [hid_t]),
            ("H5Dopen2", hid_t, [hid_t, C.c_char_p, hid_t]), ("H5Dget_space", hid_t, [hid_t]), ("H5Dget_type", hid_t, [hid_t]),
            ("H5Dread", C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            ("H5Sget_simple_extent_ndims", C.c_int, [hid_t]),
            ("H5Sget_simple_extent_dims", C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
            ("H5Acreate2", hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]), ("H5Awrite", C.c_int, [hid_t, hid_t, C.c_void_p]),
            ("H5Aclose", C.c_int, [hid_t]), ("H5Aopen", hid_t, [hid_t, C.c_char_p, hid_t]), ("H5Aread", C.c_int, [hid_t, hid_t, C.c_void_p]),
            ("H5Aget_type", hid_t, [hid_t]), ("H5Aget_space", hid_t, [hid_t]), ("H5Aexists", C.c_int, [hid_t, C.c_char_p]),
            ("H5Tcopy", hid_t, [hid_t]), ("H5Tset_size", C.c_int, [hid_t, C.c_size_t]), ("H5Tclose", C.c_int, [hid_t]),
            ("H5Tget_class", C.c_int, [hid_t]), ("H5Tget_size", C.c_size_t, [hid_t]), ("H5Tget_sign", C.c_int, [hid_t]),
            ("H5Lexists", C.c_int, [hid_t, C.c_char_p, hid_t]), ("H5Oopen", hid_t, [hid_t, C.c_char_p, hid_t]), ("H5Oclose", C.c_int, [hid_t]),
            ("H5Eset_auto2", C.c_int, [hid_t, C.c_void_p, C.c_void_p]),
        ]:
            f = getattr(l, name)
            f.restype, f.argtypes = res, args
        l.H5Eset_auto2(0, None, None)            # errors are reported through return codes below
        _lib = l
    return _lib


def _g(name):
    return hid_t.in_dll(lib(), name).value


_NP2H5 = {"u1": "H5T_NATIVE_UINT8_g", "i4": "H5T_NATIVE_INT32_g", "u4": "H5T_NATIVE_UINT32_g", "i8": "H5T_NATIVE_INT64_g",
          "u8": "H5T_NATIVE_UINT64_g", "f8": "H5T_NATIVE_DOUBLE_g", "f4": "H5T_NATIVE_FLOAT_g"}


def _chk(v, what):
    if v < 0:
        raise IOError("libhdf5: %s failed" % what)
    return v


class File(object):
    def __init__(self, path, mode="r"):
        l = lib()
        p = path.encode()
        self.id = _chk(l.H5Fcreate(p, 2, 0, 0) if mode == "w" else l.H5Fopen(p, 0, 0), "open " + path)

    def close(self):
        if self.id is not None:
            lib().H5Fclose(self.id)
            self.id = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- writing ---------------------------------------------------------------------------
    def create_group(self, path):
        g = _chk(lib().H5Gcreate2(self.id, path.encode(), 0, 0, 0), "create group " + path)
        lib().H5Gclose(g)

    def _strtype(self, n):
        t = lib().H5Tcopy(_g("H5T_C_S1_g"))
        lib().H5Tset_size(t, max(n, 1))
        return t

    def create_array(self, path, arr, compress=True):
        """Chunked, shuffled, deflate-1 dataset (PyTables ``create_carray(..., filters=Filters(1, 'zlib'))``)."""
        l = lib()
        a = np.ascontiguousarray(arr)
        if a.dtype.kind in "US":
            a = np.char.encode(a, "utf-8") if a.dtype.kind == "U" else a
            t, own = self._strtype(a.dtype.itemsize), True
        else:
            t, own = _g(_NP2H5[a.dtype.str[1:]]), False
        dims = (hsize_t * max(a.ndim, 1))(*(a.shape or (1,)))
        sp = l.H5Screate_simple(max(a.ndim, 1), dims, None)
        pl = 0
        if compress and a.size:
            pl = l.H5Pcreate(_g("H5P_CLS_DATASET_CREATE_ID_g"))
            chunk = (hsize_t * max(a.ndim, 1))(*[max(1, min(int(s), 1 << 16 if i == 0 else int(s))) for i, s in enumerate(a.shape or (1,))])
            l.H5Pset_chunk(pl, max(a.ndim, 1), chunk)
            l.H5Pset_shuffle(pl)
            l.H5Pset_deflate(pl, 1)
        d = _chk(l.H5Dcreate2(self.id, path.encode(), t, sp, 0, pl, 0), "create dataset " + path)
        if a.size:
            _chk(l.H5Dwrite(d, t, 0, 0, 0, a.ctypes.data_as(C.c_void_p)), "write " + path)
        l.H5Dclose(d)
        l.H5Sclose(sp)
        if pl:
            l.H5Pclose(pl)
        if own:
            l.H5Tclose(t)

    def set_attr(self, obj_path, name, value):
        """numpy scalars/arrays natively; ``bytes`` as a fixed string; any other Python object pickled (PyTables' rule)."""
        l = lib()
        o = _chk(l.H5Oopen(self.id, obj_path.encode(), 0), "open " + obj_path)
        try:
            if isinstance(value, (tuple, list, dict, str)):
                value = pickle.dumps(value, 0)
            if isinstance(value, bytes):
                t = self._strtype(len(value))
                sp = l.H5Screate(0)
                a = _chk(l.H5Acreate2(o, name.encode(), t, sp, 0, 0), "create attr " + name)
                buf = C.create_string_buffer(value, max(len(value), 1))
                l.H5Awrite(a, t, buf)
                l.H5Aclose(a); l.H5Sclose(sp); l.H5Tclose(t)
                return
            v = np.asarray(value)
            if v.dtype == np.bool_:
                v = v.astype(np.uint8)
            t = _g(_NP2H5[v.dtype.str[1:]])
            if v.ndim == 0:
                sp = l.H5Screate(0)
            else:
                dims = (hsize_t * v.ndim)(*v.shape)
                sp = l.H5Screate_simple(v.ndim, dims, None)
            a = _chk(l.H5Acreate2(o, name.encode(), t, sp, 0, 0), "create attr " + name)
            v = np.ascontiguousarray(v)
            l.H5Awrite(a, t, v.ctypes.data_as(C.c_void_p))
            l.H5Aclose(a); l.H5Sclose(sp)
        finally:
            l.H5Oclose(o)

    # -- reading ---------------------------------------------------------------------------
    def exists(self, path):
        parts = [p for p in path.split("/") if p]
        cur = ""
        for p in parts:
            cur += "/" + p
            if lib().H5Lexists(self.id, cur.encode(), 0) <= 0:
                return False
        return True

    @staticmethod
    def _np_type(t):
        l = lib()
        cls, size = l.H5Tget_class(t), l.H5Tget_size(t)
        if cls == 0:     # integer
            return np.dtype(("u" if l.H5Tget_sign(t) == 0 else "i") + str(size)), None
        if cls == 1:     # float
            return np.dtype("f" + str(size)), None
        if cls == 3:     # string
            return np.dtype("S" + str(size)), size
        raise IOError("unsupported HDF5 type class %d" % cls)

    def read_array(self, path):
        l = lib()
        d = _chk(l.H5Dopen2(self.id, path.encode(), 0), "open dataset " + path)
        sp, t = l.H5Dget_space(d), l.H5Dget_type(d)
        nd = l.H5Sget_simple_extent_ndims(sp)
        dims = (hsize_t * max(nd, 1))()
        l.H5Sget_simple_extent_dims(sp, dims, None)
        dt, ssize = self._np_type(t)
        out = np.empty(tuple(int(x) for x in dims[:nd]), dtype=dt)
        mt = self._strtype(ssize) if ssize else _g(_NP2H5[dt.str[1:]])
        if out.size:
            _chk(l.H5Dread(d, mt, 0, 0, 0, out.ctypes.data_as(C.c_void_p)), "read " + path)
        if ssize:
            l.H5Tclose(mt)
        l.H5Tclose(t); l.H5Sclose(sp); l.H5Dclose(d)
        return out

    def get_attr(self, obj_path, name):
        l = lib()
        o = _chk(l.H5Oopen(self.id, obj_path.encode(), 0), "open " + obj_path)
        try:
            a = _chk(l.H5Aopen(o, name.encode(), 0), "open attr " + name)
            t, sp = l.H5Aget_type(a), l.H5Aget_space(a)
            dt, ssize = self._np_type(t)
            nd = l.H5Sget_simple_extent_ndims(sp)
            dims = (hsize_t * max(nd, 1))()
            if nd:
                l.H5Sget_simple_extent_dims(sp, dims, None)
            out = np.empty(tuple(int(x) for x in dims[:nd]), dtype=dt)
            mt = self._strtype(ssize) if ssize else _g(_NP2H5[dt.str[1:]])
            l.H5Aread(a, mt, out.ctypes.data_as(C.c_void_p))
            if ssize:
                l.H5Tclose(mt)
            l.H5Tclose(t); l.H5Sclose(sp); l.H5Aclose(a)
            if ssize:
                raw = out.tobytes().rstrip(b"\x00") if out.ndim == 0 else out
                if isinstance(raw, bytes) and raw[:1] in (b"(", b"]", b"}", b"V", b"S", b"\x80"):
                    try:
                        return pickle.loads(out.tobytes())
                    except Exception:
                        return raw
                return raw
            return out[()] if out.ndim == 0 else out
        finally:
            l.H5Oclose(o)
