# -*- coding: utf-8 -*-
"""ctypes binding of ``libecb.so`` (``include/ecb.h``) -- the device side of the
reference's ``process_convert_bam`` + merge + A/N construction
(``alntools/bam_utils.py:198-363, 680-724, 768-847``).

There is no CPU fallback here on purpose: if the library is missing or no GPU is
present, constructing an :class:`EcBuilder` raises.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("ECB_LIB", "libecb.so"))   # ECB_LIB: A/B builds of the kernel

F_RANGES = 1
F_MULTISAMPLE = 2
F_VERIFY = 4
HAP_SHIFT = 16
FLAG_MATE_OTHER_REF = 0x1000
FLAG_NEXT_POS_NEG = 0x2000

#: every symbol include/ecb.h declares
SYMBOLS = ("ecb_abi_version", "ecb_device_count", "ecb_create", "ecb_destroy", "ecb_reset", "ecb_last_error",
           "ecb_push", "ecb_push_device", "ecb_hint_reads", "ecb_push_cells", "ecb_push_cells_device", "ecb_verify_device", "ecb_finalize", "ecb_export",
           "ecb_export_device", "ecb_export_ranges", "ecb_export_range_minmax", "ecb_export_pairs", "ecb_ms_filter", "ecb_ms_export",
           "ecb_export_read_ec", "ecb_table_sizes",
           "ecb_table_export_device", "ecb_table_merge_device", "ecb_table_export_parts_device",
           "ecb_table_adopt_device", "ecb_table_merge_batch_device", "ecb_table_adopt_batch_device",
           "ecb_export_firsts_device", "ecb_assemble_ranges_device", "ecb_table_rebase_device",
           "ecb_export_ec_keys_device", "ecb_ms_local_triples_device", "ecb_ms_adopt_triples_device", "ecb_counters", "ecb_add_counters", "ecb_profile",
           "ecb_profile_read", "ecb_profile_kernel", "ecb_csr_to_hapcsc_device", "ecb_hapcsc_to_csr_device", "ecb_release_scratch",
           "ecb_csr_to_hapcsc", "ecb_hapcsc_to_csr", "ecb_merge", "ecb_push_device_tiled", "ecb_verify_device_tiled")
ABI_VERSION = 4            # include/ecb.h: ECB_ABI_VERSION


class EcbError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "libecb error %d: %s" % (code, msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("n_loci", C.c_uint32),
                ("n_haplotypes", C.c_uint32), ("flags", C.c_uint32), ("reserved", C.c_uint32),
                ("ec_capacity", C.c_uint64), ("arena_capacity", C.c_uint64),
                ("max_batch_records", C.c_uint64)]


class Sizes(C.Structure):
    _fields_ = [("n_ecs", C.c_uint64), ("nnz_a", C.c_uint64), ("n_samples", C.c_uint64),
                ("nnz_n", C.c_uint64), ("all_alignments", C.c_uint64), ("valid_alignments", C.c_uint64),
                ("n_reads", C.c_uint64)]


class MsSizes(C.Structure):
    _fields_ = [("n_cells_seen", C.c_uint64), ("n_cells_kept", C.c_uint64), ("n_ecs_kept", C.c_uint64),
                ("nnz_a", C.c_uint64), ("nnz_n", C.c_uint64)]


_lib = None


def load():
    """``ctypes.CDLL`` of the in-tree library (raises if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64 (same SONAME as
    # /opt/rocm's).  Importing torch first makes the loader bind libecb's libamdhip64.so.7 to the copy
    # torch already mapped, so tensors and libecb share one runtime (two copies cannot both open the GPU).
    # A process that will never hold a tensor says so (ALNTOOLS_TORCH=0: the command line does, for its one-GPU commands --
    # everything it needs goes through host pointers) and libecb then runs on the system's HIP runtime, PyTorch not imported.
    if os.environ.get("ALNTOOLS_TORCH", "1") != "0" or "torch" in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not found: build it with `python -m alntools_amd.build` "
                          "(the HIP path has no fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.ecb_abi_version.restype = C.c_int
    ab = "ECB_LIB" in os.environ        # (an A/B build named by hand, tools/: may be an older ABI; what it lacks is simply not bound)
    if lib.ecb_abi_version() != ABI_VERSION and not ab:
        raise ImportError("%s has ABI %d, this binding is for ABI %d: rebuild it (python -m alntools_amd.build)"
                          % (LIB_PATH, lib.ecb_abi_version(), ABI_VERSION))
    vp, u64, sz = C.c_void_p, C.c_uint64, C.c_size_t
    lib.ecb_abi_version.restype = C.c_int
    lib.ecb_device_count.restype = C.c_int
    lib.ecb_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    lib.ecb_destroy.argtypes = [vp]
    lib.ecb_destroy.restype = None
    lib.ecb_reset.argtypes = [vp]
    lib.ecb_last_error.argtypes = [vp]
    lib.ecb_last_error.restype = C.c_char_p
    lib.ecb_push.argtypes = [vp, vp, vp, vp, vp, sz]
    lib.ecb_push_device.argtypes = [vp, vp, vp, vp, vp, sz]
    lib.ecb_hint_reads.argtypes = [vp, u64]
    lib.ecb_push_cells.argtypes = [vp, vp, u64, sz]
    lib.ecb_push_cells_device.argtypes = [vp, vp, u64, sz]
    lib.ecb_verify_device.argtypes = [vp, vp, vp, vp, sz, C.POINTER(u64), C.POINTER(u64)]
    lib.ecb_finalize.argtypes = [vp, C.POINTER(Sizes)]
    lib.ecb_export.argtypes = [vp] + [vp] * 6
    lib.ecb_export_device.argtypes = [vp] + [vp] * 6
    lib.ecb_export_ranges.argtypes = [vp, vp]
    lib.ecb_export_read_ec.argtypes = [vp, vp]
    lib.ecb_export_pairs.argtypes = [vp, vp, vp, vp, vp]
    lib.ecb_export_range_minmax.argtypes = [vp, vp, vp]
    lib.ecb_ms_filter.argtypes = [vp, C.c_uint32, C.c_int64, C.POINTER(MsSizes)]
    lib.ecb_ms_export.argtypes = [vp] + [vp] * 7
    lib.ecb_table_sizes.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    lib.ecb_table_export_device.argtypes = [vp, vp, vp, u64]
    lib.ecb_table_merge_device.argtypes = [vp, vp, u64, vp, u64]
    lib.ecb_table_export_parts_device.argtypes = [vp, vp, vp, u64, C.c_uint32, C.POINTER(u64), C.POINTER(u64)]
    lib.ecb_table_adopt_device.argtypes = [vp, vp, u64, vp, u64]
    lib.ecb_table_rebase_device.argtypes = [vp, vp, u64, u64]
    for f in (lib.ecb_table_merge_batch_device, lib.ecb_table_adopt_batch_device):
        f.argtypes = [vp, C.c_uint32, C.POINTER(vp), C.POINTER(u64), C.POINTER(vp), C.POINTER(u64)]
    lib.ecb_export_ec_keys_device.argtypes = [vp, vp]
    lib.ecb_export_firsts_device.argtypes = [vp, vp]
    lib.ecb_assemble_ranges_device.argtypes = [vp, C.c_uint32] + [C.POINTER(vp)] * 5 + [C.POINTER(u64)] * 2 + [u64, u64, u64, C.POINTER(Sizes)]
    lib.ecb_ms_local_triples_device.argtypes = [vp, vp, vp, vp, vp, u64, u64, vp, vp, vp, C.POINTER(u64)]
    lib.ecb_ms_adopt_triples_device.argtypes = [vp, C.c_uint32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(u64), C.POINTER(u64)]
    lib.ecb_counters.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    lib.ecb_add_counters.argtypes = [vp, u64, u64, u64]
    lib.ecb_profile.argtypes = [vp, C.c_int]
    lib.ecb_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(u64), C.POINTER(u64)]
    if not ab or hasattr(lib, "ecb_merge"):
        lib.ecb_merge.argtypes = [C.POINTER(vp), C.c_uint32, vp, C.POINTER(Sizes)]
    if not ab or hasattr(lib, "ecb_push_device_tiled"):
        lib.ecb_push_device_tiled.argtypes = [vp, vp, sz]
        lib.ecb_verify_device_tiled.argtypes = [vp, vp, sz, C.POINTER(u64), C.POINTER(u64)]
    if not ab or hasattr(lib, "ecb_profile_kernel"):
        lib.ecb_profile_kernel.argtypes = [vp]
        lib.ecb_profile_kernel.restype = C.c_char_p
    if not ab or hasattr(lib, "ecb_csr_to_hapcsc"):
        lib.ecb_csr_to_hapcsc.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp, u64, C.POINTER(u64)]
        lib.ecb_hapcsc_to_csr.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, u64, vp, vp, vp, C.POINTER(u64)]
    lib.ecb_csr_to_hapcsc_device.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp, C.POINTER(u64)]
    lib.ecb_hapcsc_to_csr_device.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, u64, vp, vp, vp, C.POINTER(u64)]
    _lib = lib
    return lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _dev_ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def tile_tuples(read_id, locus, hapflag, out=None):
    """Three device arrays of ``n`` records -> one int32 tensor of whole tiles for :meth:`EcBuilder.push_device_tiled` (tile t = words
    ``[1536 t, 1536 t + 1536)`` = 512 read ids | 512 loci | 512 haplotype/flag words; the last tile's padding is zero)."""
    import torch
    n = read_id.numel()
    nt = (n + 511) // 512
    tiles = out if out is not None else torch.zeros(nt * 1536, dtype=torch.int32, device=read_id.device)
    v = tiles[:nt * 1536].view(nt, 3, 512)
    full = n // 512
    for k, src in enumerate((read_id, locus, hapflag)):
        src = src.view(torch.int32) if src.dtype != torch.int32 else src
        if full:
            v[:full, k, :] = src[:full * 512].view(full, 512)
        if n > full * 512:
            v[full, k, :n - full * 512] = src[full * 512:]
    return tiles


def csr_to_hapcsc(indptr, indices, data, n_loci, n_haps, nnz=None):
    """f-2 on device tensors (int32, CUDA): CSR(bitmask) -> (csc_indptr [H, T+1], csc_indices [total], nnz per haplotype).
    Haplotype h's row indices are ``csc_indices[starts[h]:starts[h+1]]`` (``bin_utils.ec2emase``'s per-haplotype CSC).
    One call when the row-index buffer sized for every non-zero carrying every haplotype (``nnz`` x H) stays below 1 GiB; beyond
    that (many haplotypes, sparse masks: up to 31 x what is needed) the library is asked for the count of set bits first."""
    import torch
    lib = load()
    dev = indptr.device
    E = indptr.numel() - 1
    tot = C.c_uint64()
    nnz = indices.numel() if nnz is None else int(nnz)
    cap = nnz * n_haps
    if cap >= (1 << 28):                             # (a count first: the exact size)
        rc = lib.ecb_csr_to_hapcsc_device(dev.index or 0, E, n_loci, n_haps, _dev_ptr(indptr), _dev_ptr(indices), _dev_ptr(data),
                                          None, None, C.byref(tot))
        if rc != 0:
            raise EcbError(rc, (lib.ecb_last_error(None) or b"").decode())
        cap = tot.value
    cptr = torch.empty((n_haps, n_loci + 1), dtype=torch.int32, device=dev)
    cidx = torch.empty(max(cap, 1), dtype=torch.int32, device=dev)
    rc = lib.ecb_csr_to_hapcsc_device(dev.index or 0, E, n_loci, n_haps, _dev_ptr(indptr), _dev_ptr(indices), _dev_ptr(data),
                                      _dev_ptr(cptr), _dev_ptr(cidx), C.byref(tot))
    if rc != 0:
        raise EcbError(rc, (lib.ecb_last_error(None) or b"").decode())
    return cptr, cidx[:tot.value]


def csr_to_hapcsc_host(indptr, indices, data, n_loci, n_haps, device=0):
    """f-2 on HOST arrays (``ecb_csr_to_hapcsc``: the library holds the device buffers itself; no PyTorch involved):
    CSR(bitmask) -> (csc_indptr int32 [H, T+1], csc_indices int32 [total])."""
    lib = load()
    ip, ix, da = (np.ascontiguousarray(a, dtype=np.int32) for a in (indptr, indices, data))
    E = len(ip) - 1
    tot = C.c_uint64()
    rc = lib.ecb_csr_to_hapcsc(device, E, n_loci, n_haps, _ptr(ip), _ptr(ix), _ptr(da), None, None, 0, C.byref(tot))
    if rc != 0:
        raise EcbError(rc, (lib.ecb_last_error(None) or b"").decode())
    cptr = np.empty((n_haps, n_loci + 1), dtype=np.int32)
    cidx = np.empty(max(tot.value, 1), dtype=np.int32)
    rc = lib.ecb_csr_to_hapcsc(device, E, n_loci, n_haps, _ptr(ip), _ptr(ix), _ptr(da), _ptr(cptr), _ptr(cidx), len(cidx), C.byref(tot))
    if rc != 0:
        raise EcbError(rc, (lib.ecb_last_error(None) or b"").decode())
    return cptr, cidx[:tot.value]


def hapcsc_to_csr_host(csc_indptr, csc_indices, n_ecs, device=0):
    """f-2 inverse on HOST arrays (``ecb_hapcsc_to_csr``): per-haplotype CSC -> CSR(bitmask) (indptr, indices, data), int32."""
    lib = load()
    cptr = np.ascontiguousarray(csc_indptr, dtype=np.int32)
    cidx = np.ascontiguousarray(csc_indices, dtype=np.int32)
    H, T1 = cptr.shape
    total = len(cidx)
    ip = np.empty(n_ecs + 1, dtype=np.int32)
    ix = np.empty(max(total, 1), dtype=np.int32)
    da = np.empty(max(total, 1), dtype=np.int32)
    nnz = C.c_uint64()
    rc = lib.ecb_hapcsc_to_csr(device, n_ecs, T1 - 1, H, _ptr(cptr), _ptr(cidx), total, _ptr(ip), _ptr(ix), _ptr(da), C.byref(nnz))
    if rc != 0:
        raise EcbError(rc, (lib.ecb_last_error(None) or b"").decode())
    return ip, ix[:nnz.value], da[:nnz.value]


def hapcsc_to_csr(csc_indptr, csc_indices, n_ecs):
    """f-2 inverse on device tensors: per-haplotype CSC -> CSR(bitmask) (``bin_utils.emase2ec``: A = sum 2^h M_h)."""
    import torch
    lib = load()
    dev = csc_indptr.device
    H, T1 = csc_indptr.shape
    total = csc_indices.numel()
    ip = torch.empty(n_ecs + 1, dtype=torch.int32, device=dev)
    ix = torch.empty(total, dtype=torch.int32, device=dev)
    da = torch.empty(total, dtype=torch.int32, device=dev)
    nnz = C.c_uint64()
    rc = lib.ecb_hapcsc_to_csr_device(dev.index or 0, n_ecs, T1 - 1, H, _dev_ptr(csc_indptr), _dev_ptr(csc_indices), total,
                                      _dev_ptr(ip), _dev_ptr(ix), _dev_ptr(da), C.byref(nnz))
    if rc != 0:
        raise EcbError(rc, (lib.ecb_last_error(None) or b"").decode())
    return ip, ix[:nnz.value], da[:nnz.value]


class EcBuilder(object):
    """One handle = one GPU.  Push record tuples, finalize, read back CSR A and N."""

    def __init__(self, n_loci, n_haplotypes, device=0, track_ranges=False, ec_capacity=0,
                 arena_capacity=0, max_batch_records=0, multisample=False, verify=False):
        self._lib = load()
        self._h = C.c_void_p()
        flags = (F_RANGES if track_ranges else 0) | (F_MULTISAMPLE if multisample else 0) | (F_VERIFY if verify else 0)
        self.multisample = multisample
        cfg = Config(C.sizeof(Config), device, n_loci, n_haplotypes, flags, 0, ec_capacity,
                     arena_capacity, max_batch_records)
        rc = self._lib.ecb_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            raise EcbError(rc, (self._lib.ecb_last_error(None) or b"").decode())
        self.n_loci, self.n_haplotypes, self.track_ranges = n_loci, n_haplotypes, track_ranges
        self.sizes = None

    # -- plumbing ------------------------------------------------------------
    def _chk(self, rc):
        if rc != 0:
            raise EcbError(rc, (self._lib.ecb_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.ecb_destroy(self._h)
            self._h = None          # (not C.c_void_p(): at interpreter shutdown the ctypes module may already be gone)

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def reset(self):
        """Forget all input and results; allocations (table, arena, staging) are kept."""
        self._chk(self._lib.ecb_reset(self._h))
        self.sizes = None

    # -- input ---------------------------------------------------------------
    def push(self, read_id, locus, hapflag, pos=None):
        """Host (numpy) tuple streams; a read may straddle calls."""
        rid = np.ascontiguousarray(read_id, dtype=np.uint32)
        loc = np.ascontiguousarray(locus, dtype=np.uint32)
        hf = np.ascontiguousarray(hapflag, dtype=np.uint32)
        ps = None if pos is None else np.ascontiguousarray(pos, dtype=np.int32)
        if not (len(rid) == len(loc) == len(hf)) or (ps is not None and len(ps) != len(rid)):
            raise ValueError("tuple streams differ in length")
        self._chk(self._lib.ecb_push(self._h, _ptr(rid), _ptr(loc), _ptr(hf), _ptr(ps), len(rid)))

    def push_device(self, read_id, locus, hapflag, pos=None):
        """torch tensors already in HBM (int32/uint32 bit patterns); whole reads per call."""
        n = read_id.numel()
        for t in (read_id, locus, hapflag) + ((pos,) if pos is not None else ()):
            if not t.is_cuda or not t.is_contiguous() or t.element_size() != 4 or t.numel() != n:
                raise ValueError("device tuple streams must be contiguous 4-byte CUDA tensors of equal length")
        self._chk(self._lib.ecb_push_device(self._h, _dev_ptr(read_id), _dev_ptr(locus), _dev_ptr(hapflag),
                                            _dev_ptr(pos), n))

    def push_device_tiled(self, tiles, n):
        """``n`` records in ONE int32 CUDA tensor of whole tiles (``ecb_push_device_tiled``: 512 read ids | 512 loci | 512 haplotype/flag
        words per tile, ``ceil(n / 512) * 1536`` words); whole reads per call.  :func:`tile_tuples` lays three arrays out this way."""
        if not tiles.is_cuda or not tiles.is_contiguous() or tiles.element_size() != 4 or tiles.numel() < (n + 511) // 512 * 1536:
            raise ValueError("tiles: a contiguous 4-byte CUDA tensor of ceil(n / 512) * 1536 words")
        self._chk(self._lib.ecb_push_device_tiled(self._h, _dev_ptr(tiles), n))

    def verify_device_tiled(self, tiles, n):
        bad, skipped = C.c_uint64(), C.c_uint64()
        self._chk(self._lib.ecb_verify_device_tiled(self._h, _dev_ptr(tiles), n, C.byref(bad), C.byref(skipped)))
        return bad.value, skipped.value

    def hint_reads(self, max_reads):
        """The stream holds at most ``max_reads`` reads: ``push_device`` then waits for the device once per call, not twice
        (``ecb_hint_reads``; running past the bound is a contract error; 0 takes the bound back)."""
        self._chk(self._lib.ecb_hint_reads(self._h, int(max_reads)))

    def verify_device(self, read_id, locus, hapflag):
        """Independent exactness pass over the device-resident stream that was pushed: -> (reads whose target set differs
        from their EC's stored key, reads that took the long-read compare)."""
        bad, skipped = C.c_uint64(), C.c_uint64()
        self._chk(self._lib.ecb_verify_device(self._h, _dev_ptr(read_id), _dev_ptr(locus), _dev_ptr(hapflag),
                                              read_id.numel(), C.byref(bad), C.byref(skipped)))
        return bad.value, skipped.value

    def push_cells(self, meta, first_read):
        """Multisample: ``cell | file << 22`` of reads ``[first_read, first_read + len(meta))``."""
        m = np.ascontiguousarray(meta, dtype=np.uint32)
        self._chk(self._lib.ecb_push_cells(self._h, _ptr(m), first_read, len(m)))

    def push_cells_device(self, meta, first_read):
        """The same from an int32 CUDA tensor (kept alive by the caller until ``finalize``)."""
        self._chk(self._lib.ecb_push_cells_device(self._h, _dev_ptr(meta), first_read, meta.numel()))

    def ms_filter_sizes(self, n_cells, minimum_count):
        """``ecb_ms_filter`` alone: the result stays on the device (``ms_filter`` also copies it to the host) -> its sizes."""
        m = MsSizes()
        self._chk(self._lib.ecb_ms_filter(self._h, n_cells, minimum_count, C.byref(m)))
        return {k: int(getattr(m, k)) for k, _ in MsSizes._fields_}

    # -- results -------------------------------------------------------------
    def finalize(self):
        s = Sizes()
        self._chk(self._lib.ecb_finalize(self._h, C.byref(s)))
        self.sizes = {k: int(getattr(s, k)) for k, _ in Sizes._fields_}
        return self.sizes

    def export(self):
        """-> dict of int32 numpy arrays: indptrA, indicesA, dataA, indptrN, indicesN, dataN."""
        s = self.sizes or self.finalize()
        E, nnz, S, nnzn = s["n_ecs"], s["nnz_a"], s["n_samples"], s["nnz_n"]
        out = dict(indptrA=np.empty(E + 1, np.int32), indicesA=np.empty(nnz, np.int32),
                   dataA=np.empty(nnz, np.int32))
        if self.multisample:
            self._chk(self._lib.ecb_export(self._h, _ptr(out["indptrA"]), _ptr(out["indicesA"]), _ptr(out["dataA"]),
                                           None, None, None))
            return out
        out.update(indptrN=np.empty(S + 1, np.int32), indicesN=np.empty(nnzn, np.int32), dataN=np.empty(nnzn, np.int32))
        self._chk(self._lib.ecb_export(self._h, *[_ptr(out[k]) for k in
                                                  ("indptrA", "indicesA", "dataA", "indptrN", "indicesN", "dataN")]))
        return out

    def export_pairs(self):
        """Multisample: distinct (EC, cell, file) triples -> dict(ec, cell, file, count, first), sorted by (ec, cell, file)."""
        s = self.sizes or self.finalize()
        n = s["nnz_n"]
        ec, meta, cnt, first = (np.empty(n, np.uint32) for _ in range(4))
        self._chk(self._lib.ecb_export_pairs(self._h, _ptr(ec), _ptr(meta), _ptr(cnt), _ptr(first)))
        return dict(ec=ec.astype(np.int64), cell=(meta & ((1 << 22) - 1)).astype(np.int64), file=(meta >> 22).astype(np.int64),
                    count=cnt.astype(np.int64), first=first.astype(np.int64))

    def ms_filter(self, n_cells, minimum_count):
        """Multisample, after finalize: cell order, minimum-count filter, EC re-rank, CSC N and the surviving rows of A, on
        the device (``bam_utils_multisample.py:596-636, 737-791``) -> dict(kept_cells [cell ids in sample order], indptrA,
        indicesA, dataA, indptrN, indicesN, dataN, n_cells_seen)."""
        m = MsSizes()
        self._chk(self._lib.ecb_ms_filter(self._h, n_cells, minimum_count, C.byref(m)))
        S, E2, nnza, nnzn = int(m.n_cells_kept), int(m.n_ecs_kept), int(m.nnz_a), int(m.nnz_n)
        out = dict(kept_cells=np.empty(S, np.uint32), indptrA=np.empty(E2 + 1, np.int32), indicesA=np.empty(nnza, np.int32),
                   dataA=np.empty(nnza, np.int32), indptrN=np.empty(S + 1, np.int32), indicesN=np.empty(nnzn, np.int32),
                   dataN=np.empty(nnzn, np.int32))
        self._chk(self._lib.ecb_ms_export(self._h, *[_ptr(out[k]) for k in
                                                     ("kept_cells", "indptrA", "indicesA", "dataA", "indptrN", "indicesN", "dataN")]))
        out["n_cells_seen"] = int(m.n_cells_seen)
        return out

    def export_ranges(self):
        out = np.empty((self.n_loci, self.n_haplotypes), np.int64)
        self._chk(self._lib.ecb_export_ranges(self._h, _ptr(out)))
        return out

    def export_range_minmax(self):
        mn = np.empty((self.n_loci, self.n_haplotypes), np.int32)
        mx = np.empty((self.n_loci, self.n_haplotypes), np.int32)
        self._chk(self._lib.ecb_export_range_minmax(self._h, _ptr(mn), _ptr(mx)))
        return mn, mx

    def export_read_ec(self):
        s = self.sizes or self.finalize()
        out = np.empty(s["n_reads"], np.int32)
        self._chk(self._lib.ecb_export_read_ec(self._h, _ptr(out)))
        return out

    # -- multi-GPU -----------------------------------------------------------
    def table_sizes(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._chk(self._lib.ecb_table_sizes(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def table_export_device(self, entries, pairs, read_base):
        self._chk(self._lib.ecb_table_export_device(self._h, _dev_ptr(entries), _dev_ptr(pairs), read_base))

    def table_merge_device(self, entries, n_entries, pairs, n_pairs):
        self._chk(self._lib.ecb_table_merge_device(self._h, _dev_ptr(entries), n_entries, _dev_ptr(pairs), n_pairs))

    def table_export_parts_device(self, entries, pairs, read_base, n_parts):
        """Export grouped into ``n_parts`` key ranges -> (entry_offsets, pair_offsets), lists of n_parts + 1."""
        eo, po = (C.c_uint64 * (n_parts + 1))(), (C.c_uint64 * (n_parts + 1))()
        self._chk(self._lib.ecb_table_export_parts_device(self._h, _dev_ptr(entries), _dev_ptr(pairs), read_base, n_parts, eo, po))
        return list(eo), list(po)

    def _batch(self, fn, tables):
        """``tables``: list of (entries tensor, n_entries, pairs tensor, n_pairs)."""
        n = len(tables)
        pe, ne = (C.c_void_p * n)(*[t[0].data_ptr() for t in tables]), (C.c_uint64 * n)(*[t[1] for t in tables])
        pp, npr = (C.c_void_p * n)(*[t[2].data_ptr() for t in tables]), (C.c_uint64 * n)(*[t[3] for t in tables])
        self._chk(fn(self._h, n, pe, ne, pp, npr))

    def table_merge_batch_device(self, tables):
        self._batch(self._lib.ecb_table_merge_batch_device, tables)

    def table_adopt_batch_device(self, tables):
        self._batch(self._lib.ecb_table_adopt_batch_device, tables)

    def table_rebase_device(self, entries, n_entries, read_base):
        """Exported entries (device tensor), in place: first reads moved on by ``read_base``; ordered ahead of a merge on this handle."""
        self._chk(self._lib.ecb_table_rebase_device(self._h, _dev_ptr(entries), n_entries, read_base))

    def table_adopt_device(self, entries, n_entries, pairs, n_pairs):
        self._chk(self._lib.ecb_table_adopt_device(self._h, _dev_ptr(entries), n_entries, _dev_ptr(pairs), n_pairs))

    # -- finalize per key range ----------------------------------------------
    def export_piece_device(self, indptr, indices, data, counts, firsts):
        """After finalize, on the handle that merged one key range: CSR A, the EC counts and every EC's first read (global
        index) into int32 device tensors of E + 1 / nnz / nnz / E / E elements."""
        self._chk(self._lib.ecb_export_device(self._h, _dev_ptr(indptr), _dev_ptr(indices), _dev_ptr(data), None, None, _dev_ptr(counts)))
        self._chk(self._lib.ecb_export_firsts_device(self._h, _dev_ptr(firsts)))

    def assemble_ranges_device(self, pieces, n_reads, all_alignments, valid_alignments):
        """``pieces``: [(indptr, indices, data, counts, firsts, n_ecs, nnz), ...] of int32 device tensors, one per key range,
        into this (empty) handle, which is finalized afterwards -> sizes."""
        k = len(pieces)
        ptrs = [(C.c_void_p * k)(*[p[i].data_ptr() for p in pieces]) for i in range(5)]
        ne, nz = (C.c_uint64 * k)(*[p[5] for p in pieces]), (C.c_uint64 * k)(*[p[6] for p in pieces])
        s = Sizes()
        self._chk(self._lib.ecb_assemble_ranges_device(self._h, k, *ptrs, ne, nz, n_reads, all_alignments, valid_alignments, C.byref(s)))
        self.sizes = {k_: int(getattr(s, k_)) for k_, _ in Sizes._fields_}
        return self.sizes

    # -- multisample across GPUs ----------------------------------------------
    def export_ec_keys_device(self, keys):
        """After finalize: the 8-byte set hash of every EC in rank order into ``keys`` (int64 tensor of n_ecs)."""
        self._chk(self._lib.ecb_export_ec_keys_device(self._h, _dev_ptr(keys)))

    def export_device(self, indptr, indices, data):
        """After finalize: CSR A into int32 device tensors of E + 1 / nnz / nnz elements."""
        self._chk(self._lib.ecb_export_device(self._h, _dev_ptr(indptr), _dev_ptr(indices), _dev_ptr(data), None, None, None))

    def ms_local_triples_device(self, keys, indptr, indices, data, n_ecs, read_base, out_key, out_count, out_first):
        """A shard's (EC, cell, file) triples with global EC ids (its ECs found in the merged result by hash ``keys`` and
        then by key against the merged CSR rows) -> number of triples written."""
        n = C.c_uint64()
        self._chk(self._lib.ecb_ms_local_triples_device(self._h, _dev_ptr(keys), _dev_ptr(indptr), _dev_ptr(indices), _dev_ptr(data),
                                                        n_ecs, read_base, _dev_ptr(out_key), _dev_ptr(out_count), _dev_ptr(out_first),
                                                        C.byref(n)))
        return n.value

    def ms_adopt_triples_device(self, tables):
        """``tables``: [(key int64 tensor, count int32 tensor, first int32 tensor, n), ...] -> number of distinct triples."""
        k = len(tables)
        pk, pc = (C.c_void_p * k)(*[t[0].data_ptr() for t in tables]), (C.c_void_p * k)(*[t[1].data_ptr() for t in tables])
        pf, nn = (C.c_void_p * k)(*[t[2].data_ptr() for t in tables]), (C.c_uint64 * k)(*[t[3] for t in tables])
        out = C.c_uint64()
        self._chk(self._lib.ecb_ms_adopt_triples_device(self._h, k, pk, pc, pf, nn, C.byref(out)))
        if self.sizes is not None:
            self.sizes["nnz_n"] = out.value
        return out.value

    def counters(self):
        """-> (all_alignments, valid_alignments, n_reads) so far."""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._chk(self._lib.ecb_counters(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def add_counters(self, all_alignments, valid_alignments, n_reads):
        self._chk(self._lib.ecb_add_counters(self._h, all_alignments, valid_alignments, n_reads))

    # -- measurement ---------------------------------------------------------
    def profile(self, enable=True):
        self._chk(self._lib.ecb_profile(self._h, 1 if enable else 0))

    def profile_read(self):
        ms, n, r = C.c_double(), C.c_uint64(), C.c_uint64()
        self._chk(self._lib.ecb_profile_read(self._h, C.byref(ms), C.byref(n), C.byref(r)))
        return ms.value, n.value, r.value

    def merge_from(self, shards):
        """``ecb_merge``: this (empty) handle becomes the finalized result of the contiguous read shards held by the builders in
        ``shards`` (each on its own GPU, pushed but not finalized): the multi-GPU merge inside libecb, one process, peer copies."""
        k = len(shards)
        hs = (C.c_void_p * k)(*[b._h for b in shards])
        s = Sizes()
        self._chk(self._lib.ecb_merge(hs, k, self._h, C.byref(s)))
        self.sizes = {k_: int(getattr(s, k_)) for k_, _ in Sizes._fields_}
        return self.sizes

    def profile_kernel(self):
        """Name of the stream kernel the last batch launched (as rocprofv3 prints it)."""
        return (self._lib.ecb_profile_kernel(self._h) or b"").decode() if hasattr(self._lib, "ecb_profile_kernel") else ""
