# -*- coding: utf-8 -*-
"""ctypes binding of ``libbamdec.so`` (``csrc/bamdec.c``): the BAM decoder of the host side.

The reference iterates a ``pysam.AlignmentFile`` record by record in Python (``bam_utils.py:253-320``); here the BGZF blocks are
inflated by a pool of threads and the records parsed in C, and what comes back are column arrays: the raw fields the tuple
encoder needs, whether a record passes the reference's filter (``:264-270``) and whether it starts a new read (runs of equal,
space-trimmed names among the records that pass, ``:289-320``).  ``alntools_amd.bamio.BamReader`` is the pure-Python reader of
the same records (it also yields the names, which the multisample path needs); ``tests/test_host_logic.py`` holds the two to
each other.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbamdec.so")
SRC = os.path.join(_HERE, "csrc", "bamdec.c")
SYMBOLS = ("bd_abi_version", "bd_open", "bd_close", "bd_last_error", "bd_n_references", "bd_reference_name",
           "bd_reference_length", "bd_header_text", "bd_references", "bd_read", "bd_read_tuples", "bd_read_ms", "bd_ms_cells", "bd_progress")
ABI_VERSION = 4            # include/bamdec.h: bd_abi_version()
_lib = None


def build(force=False):
    """gcc + zlib, in-tree (``python -m alntools_amd.build`` calls this too)."""
    import subprocess
    import tempfile
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= os.path.getmtime(SRC):
        return LIB_PATH
    # into a file of its own, then moved into place in one step: several ranks of one run may get here at the same moment, and a
    # sibling may be mapping the library while another writes it
    fd, tmp = tempfile.mkstemp(prefix=".libbamdec.", suffix=".so", dir=_HERE)
    os.close(fd)
    try:
        subprocess.check_call([os.environ.get("CC", "gcc"), "-O2", "-Wall", "-shared", "-fPIC", "-o", tmp, SRC, "-lz", "-lpthread"])
        os.chmod(tmp, 0o755)
        os.replace(tmp, LIB_PATH)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB_PATH


def available():
    return os.path.exists(LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        if os.path.exists(SRC) and (not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(SRC)):
            build()                                  # (a library older than its source: the argument lists below may not be its own)
        l = C.CDLL(LIB_PATH)
        l.bd_abi_version.restype = C.c_int
        if l.bd_abi_version() != ABI_VERSION:
            raise ImportError("%s has ABI %d, this binding is for ABI %d: rebuild it (python -m alntools_amd.build)"
                              % (LIB_PATH, l.bd_abi_version(), ABI_VERSION))
        l.bd_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
        l.bd_close.argtypes = [C.c_void_p]
        l.bd_close.restype = None
        l.bd_last_error.argtypes = [C.c_void_p]
        l.bd_last_error.restype = C.c_char_p
        l.bd_n_references.argtypes = [C.c_void_p]
        l.bd_reference_name.argtypes = [C.c_void_p, C.c_int32]
        l.bd_reference_name.restype = C.c_char_p
        l.bd_reference_length.argtypes = [C.c_void_p, C.c_int32]
        l.bd_header_text.argtypes = [C.c_void_p]
        l.bd_header_text.restype = C.c_char_p
        l.bd_references.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.POINTER(C.c_int32))]
        l.bd_read.argtypes = [C.c_void_p, C.c_size_t, C.c_int] + [C.c_void_p] * 7 + [C.POINTER(C.c_size_t)]
        l.bd_read_tuples.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_uint32)] + \
            [C.c_void_p] * 4 + [C.POINTER(C.c_size_t)] * 2
        l.bd_read_ms.argtypes = [C.c_void_p, C.c_size_t] + [C.c_void_p] * 7 + [C.POINTER(C.c_size_t)]
        l.bd_ms_cells.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_size_t)]
        l.bd_progress.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        _lib = l
    return _lib


class NativeBamReader(object):
    """Same surface as ``bamio.BamReader`` where the single-sample path needs it (``references``, ``lengths``, ``close``);
    ``read_decoded`` instead of ``read_batch``: no names cross into Python."""

    def __init__(self, path, threads=None, trim=True):
        self._l = lib()
        self._h = C.c_void_p()
        threads = threads or int(os.environ.get("ALNTOOLS_DECODE_THREADS", min(os.cpu_count() or 1, 16)))
        rc = self._l.bd_open(os.fsencode(path), threads, C.byref(self._h))
        if rc != 0:
            raise (IOError if rc == -1 else ValueError)("%s: cannot read as BAM (bamdec error %d)" % (path, rc))
        n = self._l.bd_n_references(self._h)
        blob, blob_len, lens = C.c_void_p(), C.c_size_t(0), C.POINTER(C.c_int32)()
        if self._l.bd_references(self._h, C.byref(blob), C.byref(blob_len), C.byref(lens)) != 0:
            raise MemoryError("bamdec: %s" % self._l.bd_last_error(self._h).decode())
        # (one call and one split instead of two ctypes calls per reference: a transcriptome header has 10^5 .. 10^6 of them)
        self.references = tuple(C.string_at(blob, blob_len.value).decode("utf-8").split("\0")[:n]) if n else ()
        self.lengths = tuple(np.ctypeslib.as_array(lens, shape=(n,)).tolist()) if n else ()
        self.text = (self._l.bd_header_text(self._h) or b"").decode("utf-8", "replace")
        self.trim = 1 if trim else 0
        self._cap = 0

    def _arrays(self, n):
        if n > self._cap:
            self._flag = np.empty(n, np.uint16)
            self._i32 = [np.empty(n, np.int32) for _ in range(4)]
            self._u8 = [np.empty(n, np.uint8) for _ in range(2)]
            self._cap = n

    def read_decoded(self, max_records):
        """-> dict(flag u16, tid, pos, next_tid, next_pos i32, valid u8, head u8) of up to ``max_records`` records, or ``None``
        at the end of the file.  The arrays are views of buffers the next call overwrites."""
        self._arrays(max_records)
        n = C.c_size_t(0)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        rc = self._l.bd_read(self._h, max_records, self.trim, p(self._flag), p(self._i32[0]), p(self._i32[1]), p(self._i32[2]),
                             p(self._i32[3]), p(self._u8[0]), p(self._u8[1]), C.byref(n))
        if rc != 0:
            raise ValueError("BAM decode failed: %s" % self._l.bd_last_error(self._h).decode())
        k = n.value
        if k == 0:
            return None
        return dict(flag=self._flag[:k], tid=self._i32[0][:k], pos=self._i32[1][:k], next_tid=self._i32[2][:k],
                    next_pos=self._i32[3][:k], valid=self._u8[0][:k], head=self._u8[1][:k])

    def read_tuples(self, max_records, enc):
        """-> the tuples ``enc.encode_decoded(**self.read_decoded(max_records))`` would return (fresh arrays), made by the decoder
        itself (``bd_read_tuples``), or ``None`` at the end of the file.  ``enc``: the :class:`tuples.TupleEncoder` whose maps
        are used and whose read counter is carried on."""
        m = enc.maps
        if getattr(self, "_maps_of", None) is not m:        # (contiguous uint32 tables, made once per encoder)
            self._t2l = np.ascontiguousarray(m.tid2locus, dtype=np.uint32)
            self._t2h = np.ascontiguousarray(m.tid2hap, dtype=np.uint32)
            self._maps_of = m
        out = dict(read_id=np.empty(max_records, np.uint32), locus=np.empty(max_records, np.uint32),
                   hapflag=np.empty(max_records, np.uint32), pos=np.empty(max_records, np.int32))
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        cur, n, nv = C.c_uint32(enc.cur), C.c_size_t(0), C.c_size_t(0)
        rc = self._l.bd_read_tuples(self._h, max_records, self.trim, p(self._t2l), p(self._t2h), len(self._t2l), C.byref(cur),
                                    p(out["read_id"]), p(out["locus"]), p(out["hapflag"]), p(out["pos"]), C.byref(n), C.byref(nv))
        if rc != 0:
            raise ValueError("BAM decode failed: %s" % self._l.bd_last_error(self._h).decode())
        k = n.value
        if k == 0:
            return None
        enc.cur = cur.value
        out = {key: a[:k] for key, a in out.items()}
        out["n_valid"] = nv.value
        return out

    def read_ms(self, max_records):
        """The multisample path's scan (``bam_utils_multisample.py:257-300``): as ``read_decoded`` with ``newrun`` (that path's run
        rule: the tracked name is cut at its first space only until the first switch) instead of ``head``, plus the cell
        barcodes (field 14 of the tracked name split at ``'|||'``) of the runs started, as a list of ``str``."""
        self._arrays(max_records)
        n = C.c_size_t(0)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        rc = self._l.bd_read_ms(self._h, max_records, p(self._flag), p(self._i32[0]), p(self._i32[1]), p(self._i32[2]),
                                p(self._i32[3]), p(self._u8[0]), p(self._u8[1]), C.byref(n))
        if rc != 0:
            raise ValueError("BAM decode failed: %s" % self._l.bd_last_error(self._h).decode())
        k = n.value
        if k == 0:
            return None, []
        bytes_p, off_p, nc = C.POINTER(C.c_uint8)(), C.POINTER(C.c_uint32)(), C.c_size_t(0)
        self._l.bd_ms_cells(self._h, C.byref(bytes_p), C.byref(off_p), C.byref(nc))
        cells = []
        if nc.value:
            off = np.ctypeslib.as_array(off_p, shape=(nc.value + 1,))
            raw = bytes(np.ctypeslib.as_array(bytes_p, shape=(int(off[-1]),))) if off[-1] else b""
            cells = [raw[off[i]:off[i + 1]].decode("utf-8") for i in range(nc.value)]
        return dict(flag=self._flag[:k], tid=self._i32[0][:k], pos=self._i32[1][:k], next_tid=self._i32[2][:k],
                    next_pos=self._i32[3][:k], valid=self._u8[0][:k], newrun=self._u8[1][:k]), cells

    def progress(self):
        """Fraction of the file the records handed out so far reach (compressed bytes; what is read ahead but not parsed yet is not
        counted): monotone, 1.0 at the end."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        if self._l.bd_progress(self._h, C.byref(a), C.byref(b)) != 0 or not b.value:
            return 0.0
        return min(1.0, a.value / float(b.value))

    def close(self):
        if self._h:
            self._l.bd_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
