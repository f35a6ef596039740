# -*- coding: utf-8 -*-
"""Minimal BGZF / BAM reader and writer (zlib only).

The reference decodes BAM through pysam/htslib (``bam_utils.py:253-259``); that
stays on the host in this build too.  pysam is used when it is importable;
otherwise this module supplies the handful of raw BAM fields the hot path
consumes (``flag, refID, pos, next_refID, next_pos, read_name`` -- the fields
behind the pysam attributes read at ``bam_utils.py:264-301``).

The writer exists so that tests and the synthetic-workload generator can make
real ``.bam`` files without samtools.  It writes the 16-byte BGZF block header
and the 28-byte EOF block byte-for-byte as htslib does (the reference checks
both literally, ``bam_utils.py:29-30,134-154``).
"""
from __future__ import annotations

import os
import struct
import zlib

import numpy as np

BGZF_HEADER = b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00\x42\x43\x02\x00"
BGZF_EOF = (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00"
            b"\x1b\x00\x03\x00\x00\x00\x00\x00\x00\x00\x00\x00")
_MAX_BLOCK = 0xFF00  # uncompressed payload per BGZF block (htslib's choice)

_REC_FIXED = struct.Struct("<iiiBBHHHiiii")  # block_size .. tlen (36 bytes)


# --------------------------------------------------------------------------- #
# BGZF
# --------------------------------------------------------------------------- #
def bgzf_block(data: bytes, level: int = 6) -> bytes:
    """One complete BGZF block holding ``data`` (<= 64 KiB uncompressed)."""
    if len(data) > 0x10000:
        raise ValueError("BGZF block payload too large")
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    bsize = len(comp) + 25  # header 18 incl. BSIZE field, trailer 8, minus 1
    if bsize > 0xFFFF:
        # incompressible: store
        c = zlib.compressobj(0, zlib.DEFLATED, -15)
        comp = c.compress(data) + c.flush()
        bsize = len(comp) + 25
    return (BGZF_HEADER + struct.pack("<H", bsize) + comp +
            struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


class BgzfWriter(object):
    def __init__(self, path, mode="wb", level=6):
        self._fh = open(path, mode)
        self._buf = bytearray()
        self._level = level

    def write(self, data):
        self._buf += data
        while len(self._buf) >= _MAX_BLOCK:
            self._fh.write(bgzf_block(bytes(self._buf[:_MAX_BLOCK]), self._level))
            del self._buf[:_MAX_BLOCK]

    def flush_block(self):
        """End the current BGZF block here (next write starts a new block)."""
        if self._buf:
            self._fh.write(bgzf_block(bytes(self._buf), self._level))
            self._buf = bytearray()

    def close(self, eof=True):
        self.flush_block()
        if eof:
            self._fh.write(BGZF_EOF)
        self._fh.close()


def iter_bgzf_blocks(fh):
    """Yield ``(file_offset, payload_bytes)`` for every BGZF block of ``fh``."""
    while True:
        start = fh.tell()
        head = fh.read(12)
        if not head:
            return
        if len(head) < 12 or head[:4] != b"\x1f\x8b\x08\x04":
            raise ValueError("not a BGZF block at offset %d" % start)
        xlen = struct.unpack_from("<H", head, 10)[0]
        extra = fh.read(xlen)
        bsize = None
        off = 0
        while off < xlen:
            si = extra[off:off + 2]
            slen = struct.unpack_from("<H", extra, off + 2)[0]
            if si == b"BC":
                bsize = struct.unpack_from("<H", extra, off + 4)[0] + 1
            off += 4 + slen
        if bsize is None:
            raise ValueError("BGZF block without BC subfield at %d" % start)
        comp = fh.read(bsize - 12 - xlen - 8)
        crc, isize = struct.unpack("<II", fh.read(8))
        data = zlib.decompress(comp, -15) if isize else b""
        if len(data) != isize:
            raise ValueError("BGZF block size mismatch at %d" % start)
        yield start, data


# --------------------------------------------------------------------------- #
# BAM writer
# --------------------------------------------------------------------------- #
def _reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def pack_record(qname, flag, ref_id, pos, next_ref_id=-1, next_pos=-1, mapq=255, tlen=0):
    """One BAM alignment record with an empty CIGAR/SEQ/QUAL (allowed by the spec)."""
    name = qname.encode("utf-8") + b"\x00"
    if len(name) > 255:
        raise ValueError("read name too long for BAM")
    body = struct.pack("<iiBBHHHiiii", ref_id, pos, len(name), mapq,
                       _reg2bin(max(pos, 0), max(pos, 0) + 1), 0, flag & 0xFFFF, 0,
                       next_ref_id, next_pos, tlen) + name
    return struct.pack("<i", len(body)) + body


def write_bam(path, references, records, header_text=None, level=6):
    """Write a BAM file.

    ``references``: sequence of ``(name, length)``; ``records``: iterable of
    ``(qname, flag, ref_id, pos, next_ref_id, next_pos)``.  The header ends on a
    BGZF block boundary, as samtools writes it (the reference's single-chunk path
    relies on that, ``bam_utils.py:87-99,172,184``).
    """
    if header_text is None:
        header_text = "@HD\tVN:1.0\tSO:unsorted\n" + "".join(
            "@SQ\tSN:%s\tLN:%d\n" % (n, l) for n, l in references)
    text = header_text.encode("utf-8")
    w = BgzfWriter(path, level=level)
    hdr = bytearray(b"BAM\x01" + struct.pack("<i", len(text)) + text +
                    struct.pack("<i", len(references)))
    for n, l in references:
        nb = n.encode("utf-8") + b"\x00"
        hdr += struct.pack("<i", len(nb)) + nb + struct.pack("<i", l)
    w.write(bytes(hdr))
    w.flush_block()
    for r in records:
        w.write(pack_record(*r))
    w.close()


# --------------------------------------------------------------------------- #
# BAM reader
# --------------------------------------------------------------------------- #
class BamReader(object):
    """Sequential BAM reader exposing the raw fields of each record.

    ``references`` / ``lengths`` mirror ``pysam.AlignmentFile.references/.lengths``
    (used at ``bam_utils.py:582,615``).
    """

    def __init__(self, path):
        self.path = path
        self._fh = open(path, "rb")
        self._size = os.path.getsize(path)
        self._blocks = iter_bgzf_blocks(self._fh)
        self._buf = b""
        self._pos = 0
        self._read_header()

    # -- byte supply ---------------------------------------------------------
    def _need(self, n):
        while len(self._buf) - self._pos < n:
            try:
                _, data = next(self._blocks)
            except StopIteration:
                return False
            self._buf = self._buf[self._pos:] + data
            self._pos = 0
        return True

    def _take(self, n):
        if not self._need(n):
            raise EOFError("truncated BAM")
        b = self._buf[self._pos:self._pos + n]
        self._pos += n
        return b

    def _read_header(self):
        if self._take(4) != b"BAM\x01":
            raise ValueError("%s is not a BAM file" % self.path)
        l_text = struct.unpack("<i", self._take(4))[0]
        self.text = self._take(l_text).decode("utf-8", "replace")
        n_ref = struct.unpack("<i", self._take(4))[0]
        names, lens = [], []
        for _ in range(n_ref):
            l_name = struct.unpack("<i", self._take(4))[0]
            names.append(self._take(l_name)[:-1].decode("utf-8"))
            lens.append(struct.unpack("<i", self._take(4))[0])
        self.references = tuple(names)
        self.lengths = tuple(lens)

    # -- records -------------------------------------------------------------
    def __iter__(self):
        return self

    def __next__(self):
        """-> ``(qname, flag, ref_id, pos, next_ref_id, next_pos)``"""
        if not self._need(4):
            raise StopIteration
        (bs,) = struct.unpack_from("<i", self._buf, self._pos)
        if not self._need(4 + bs):
            raise EOFError("truncated BAM record")
        p = self._pos
        (_, ref_id, pos, l_name, _mq, _bin, _nc, flag, _ls, nref, npos,
         _tl) = _REC_FIXED.unpack_from(self._buf, p)
        qname = self._buf[p + 36:p + 36 + l_name - 1].decode("utf-8")
        self._pos = p + 4 + bs
        return qname, flag, ref_id, pos, nref, npos

    def progress(self):
        """Fraction of the file's (compressed) bytes taken in so far."""
        return min(1.0, self._fh.tell() / float(self._size)) if self._size else 0.0

    def read_batch(self, max_records):
        """Up to ``max_records`` records as column arrays
        ``(qnames list, flag u16, ref_id i32, pos i32, next_ref_id i32, next_pos i32)``."""
        q, cols = [], []
        for _ in range(max_records):
            try:
                r = self.__next__()
            except StopIteration:
                break
            q.append(r[0])
            cols.append(r[1:])
        a = np.asarray(cols, dtype=np.int64).reshape(-1, 5)
        return (q, a[:, 0].astype(np.uint16), a[:, 1].astype(np.int32), a[:, 2].astype(np.int32),
                a[:, 3].astype(np.int32), a[:, 4].astype(np.int32))

    def close(self):
        self._fh.close()
