# -*- coding: utf-8 -*-
"""ORACLE -- test infrastructure.  ctypes loader of the C restatement (oracle/ec_oracle.c)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libec_oracle.so")


class _Result(C.Structure):
    _fields_ = [("n_ecs", C.c_uint64), ("nnz", C.c_uint64), ("n_all", C.c_uint64), ("n_valid", C.c_uint64),
                ("n_reads", C.c_uint64), ("indptr", C.POINTER(C.c_int32)), ("indices", C.POINTER(C.c_int32)),
                ("data", C.POINTER(C.c_int32)), ("count", C.POINTER(C.c_int32))]


_lib = None


def load():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "ec_oracle.c")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        _lib = C.CDLL(_SO)
        _lib.ec_oracle_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int,
                                       C.POINTER(_Result)]
        _lib.ec_oracle_free.argtypes = [C.POINTER(_Result)]
    return _lib


def ec_from_tuples(read_id, locus, hapflag, n_haps, threads=1):
    """Same contract as ``ec_oracle.ec_from_tuples`` (without ranges), in C, over ``threads`` shards."""
    lib = load()
    rid = np.ascontiguousarray(read_id, dtype=np.uint32)
    loc = np.ascontiguousarray(locus, dtype=np.uint32)
    hf = np.ascontiguousarray(hapflag, dtype=np.uint32)
    r = _Result()
    rc = lib.ec_oracle_run(rid.ctypes.data, loc.ctypes.data, hf.ctypes.data, len(rid), n_haps, threads, C.byref(r))
    if rc != 0:
        raise ValueError("no valid alignments")
    try:
        E, nnz = r.n_ecs, r.nnz
        return dict(indptr=np.ctypeslib.as_array(r.indptr, (E + 1,)).copy(),
                    indices=np.ctypeslib.as_array(r.indices, (max(nnz, 1),))[:nnz].copy(),
                    data=np.ctypeslib.as_array(r.data, (max(nnz, 1),))[:nnz].copy(),
                    count=np.ctypeslib.as_array(r.count, (E,)).copy(),
                    n_all=int(r.n_all), n_valid=int(r.n_valid), n_reads=int(r.n_reads))
    finally:
        lib.ec_oracle_free(C.byref(r))
