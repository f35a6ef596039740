# -*- coding: utf-8 -*-
"""EC ``.bin`` format 2, byte-exact with the reference's ``bin_utils.ecsave2`` / ``ecload``
(``alntools/bin_utils.py:105-277, 32-102``).

The reference packs every array with ``struct.pack('<Ni', *array)`` (one Python int per
element); here the same little-endian int32 bytes come from ``ndarray.astype('<i4').tobytes()``.
Layout: ``2``; ``H`` then H x (len, utf-8 name); ``T`` then T x (len, name, H lengths);
``S`` then S x (len, name); A as CSR (``len(indptr)``, ``nnz``, indptr, indices = locus,
data = haplotype bitmask); N as CSC (``len(indptr)``, ``nnz``, indptr, indices = EC, data = count).
"""
from __future__ import annotations

from struct import pack, unpack_from

import numpy as np


class ECMatrices(object):
    """What ``ecsave2`` consumes and ``ecload`` returns: names, lengths, CSR A and CSC N.

    The slice of the reference's ``AlignmentPropertyMatrix`` (shape-constructed, finalized;
    ``AlignmentPropertyMatrix.py:148-171``, ``Sparse3DMatrix.py:189-193``) that the hot path fills.
    """

    def __init__(self, hname, lname, lengths, sname, indptrA, indicesA, dataA, indptrN, indicesN, dataN):
        self.hname, self.lname, self.sname = list(hname), list(lname), list(sname)
        self.lengths = np.asarray(lengths)
        self.indptrA, self.indicesA, self.dataA = (np.asarray(x, dtype=np.int32) for x in (indptrA, indicesA, dataA))
        self.indptrN, self.indicesN, self.dataN = (np.asarray(x, dtype=np.int32) for x in (indptrN, indicesN, dataN))

    num_haplotypes = property(lambda s: len(s.hname))
    num_loci = property(lambda s: len(s.lname))
    num_samples = property(lambda s: len(s.sname))
    num_reads = property(lambda s: len(s.indptrA) - 1)
    shape = property(lambda s: (len(s.lname), len(s.hname), len(s.indptrA) - 1))

    def haplotype_csc(self, h):
        """Per-haplotype CSC (E x T) incidence, as ``apm.data[h]`` after ``finalize()``."""
        from scipy.sparse import csr_matrix
        bit = ((self.dataA >> h) & 1).astype(np.float64)
        m = csr_matrix((bit, self.indicesA.copy(), self.indptrA.copy()), shape=(self.num_reads, self.num_loci))
        m.eliminate_zeros()
        return m.tocsc()


def _name(s):
    return pack('<i', len(s)) + pack('<{}s'.format(len(s)), s.encode('utf-8'))


def _i32(a):
    return np.ascontiguousarray(a).astype('<i4').tobytes()


def ecsave2_bytes(m):
    out = [pack('<i', 2), pack('<i', m.num_haplotypes)]
    out += [_name(h) for h in m.hname]
    out.append(pack('<i', m.num_loci))
    lens = np.asarray(m.lengths).astype(int)                       # bin_utils.py:153
    for t, name in enumerate(m.lname):
        out.append(_name(name))
        out.append(_i32(lens[t, :m.num_haplotypes]))
    out.append(pack('<i', m.num_samples))
    out += [_name(s) for s in m.sname]
    out += [pack('<i', len(m.indptrA)), pack('<i', len(m.indicesA)), _i32(m.indptrA), _i32(m.indicesA), _i32(m.dataA)]
    out += [pack('<i', len(m.indptrN)), pack('<i', len(m.indicesN)), _i32(m.indptrN), _i32(m.indicesN), _i32(m.dataN)]
    return b"".join(out)


def ecsave2(ec_filename, m):
    """Write ``m`` (:class:`ECMatrices`) as EC format 2 -- same bytes as the reference writer."""
    with open(ec_filename, 'wb') as f:
        f.write(ecsave2_bytes(m))


def ecload(ec_filename):
    """Read EC format 2 -> :class:`ECMatrices` (``bin_utils.py:32-102``)."""
    with open(ec_filename, 'rb') as f:
        b = f.read()
    o = [0]

    def i32(n=1):
        v = np.frombuffer(b, dtype='<i4', count=n, offset=o[0])
        o[0] += 4 * n
        return v

    def name():
        n = int(i32()[0])
        s = unpack_from('<{}s'.format(n), b, o[0])[0].decode('utf-8')
        o[0] += n
        return s

    fmt = int(i32()[0])
    if fmt == 1:
        raise NotImplementedError
    if fmt != 2:
        raise TypeError('Format 0 is not supported anymore.')
    H = int(i32()[0])
    hname = [name() for _ in range(H)]
    T = int(i32()[0])
    lname, lengths = [], np.zeros((T, H), dtype=float)
    for t in range(T):
        lname.append(name())
        lengths[t] = i32(H)
    S = int(i32()[0])
    sname = [name() for _ in range(S)]
    na, nnz = int(i32()[0]), int(i32()[0])
    A = (i32(na).copy(), i32(nnz).copy(), i32(nnz).copy())
    nn, nnzn = int(i32()[0]), int(i32()[0])
    N = (i32(nn).copy(), i32(nnzn).copy(), i32(nnzn).copy())
    return ECMatrices(hname, lname, lengths, sname, A[0], A[1], A[2], N[0], N[1], N[2])


def ec2emase(ec_file, emase_file, **converters):
    """``.bin`` -> EMASE ``.h5`` (``bin_utils.py:979-995``); the CSR -> per-haplotype CSC conversion runs on the GPU
    (``emase_h5.device_hapcsc``; tests may pass ``hapcsc=`` their checker)."""
    from . import emase_h5
    emase_h5.save(emase_file, ecload(ec_file), title='Converted from {}'.format(ec_file), incidence_only=False, **converters)


def emase2ec(emase_file, ec_file, **converters):
    """EMASE ``.h5`` -> ``.bin`` (``bin_utils.py:998-1028``); ``A = sum_h 2^h M_h`` is built on the GPU
    (``emase_h5.device_csr``; tests may pass ``csr=`` their checker)."""
    from . import emase_h5
    ecsave2(ec_file, emase_h5.load(emase_file, **converters))
