# -*- coding: utf-8 -*-
"""Host side of the record-tuple contract of ``include/ecb.h``: BAM header -> index maps, decoded
records -> ``(read_id, locus, hapflag, pos)`` streams.

What stays on the host is what only the host can know: names.  The reference derives targets and
haplotypes from ``@SQ`` names (``bam_utils.py:561-633``) and groups records into reads by comparing
space-trimmed query names of consecutive *valid* records (``bam_utils.py:289-320``).
"""
from __future__ import annotations

import numpy as np

from .ecb import FLAG_MATE_OTHER_REF, FLAG_NEXT_POS_NEG, HAP_SHIFT


def split_reference_name(name):
    """Split at the last ``'_'`` if its index is > 0 (``bam_utils.py:584-591``)."""
    i = name.rfind('_')
    if i > 0:
        return name[:i], name[i + 1:]
    return name, ''


class HeaderMaps(object):
    """``tid -> (locus, haplotype)`` and the names/lengths the writers need (``bam_utils.py:561-633``)."""

    def __init__(self, references, lengths, target_ids=None):
        references = list(references)
        parts = [split_reference_name(n) for n in references]
        targets = [p[0] for p in parts]
        haps = [p[1] for p in parts]
        # target file first (:571-579), then header order of first appearance (:596-598); dicts keep insertion order
        main_targets = dict.fromkeys(target_ids or [])
        main_targets.update(dict.fromkeys(targets))
        self.main_targets = list(main_targets.keys())
        self.haplotypes = sorted(set(haps))                          # '' sorts first (:602)
        t_idx = {t: i for i, t in enumerate(self.main_targets)}
        h_idx = {h: i for i, h in enumerate(self.haplotypes)}
        T, H = len(self.main_targets), len(self.haplotypes)
        # The reference tests "gettid(target_hap) in key" per (EC, target, haplotype) (:754, :801-809); that equals "this tid is
        # in the key" only when names and (target, haplotype) correspond one to one: the name rebuilt from its parts must be the
        # name, and no name may appear twice.  (A name splits back into itself unless it ends in '_' behind a non-empty target.)
        first_tid = {}
        for tid, name in enumerate(references):
            first_tid.setdefault(name, tid)
        if len(first_tid) != len(references) or any(len(h) == 0 and len(t) != len(n) for n, (t, h) in zip(references, parts)):
            for tid, (target, hap) in enumerate(parts):
                rebuilt = target if len(hap) == 0 else '{}_{}'.format(target, hap)
                if first_tid.get(rebuilt, -1) != tid:
                    raise ValueError("reference name %r does not round-trip through (target=%r, haplotype=%r): "
                                     "such headers (duplicate names, or a trailing '_') are not supported" %
                                     (references[tid], target, hap))
        self.tid2locus = np.fromiter((t_idx[t] for t in targets), dtype=np.uint32, count=len(parts))
        self.tid2hap = np.fromiter((h_idx[h] for h in haps), dtype=np.uint32, count=len(parts))
        self.lengths = np.zeros((T, H), dtype=np.int32)              # :605
        self.lengths[self.tid2locus, self.tid2hap] = np.asarray(lengths, dtype=np.int32)   # :615-633
        self.slot2tid = np.full(T * H, -1, dtype=np.int64)           # gettid(target_hap), -1 if absent (:754, :809)
        self.slot2tid[self.tid2locus.astype(np.int64) * H + self.tid2hap] = np.arange(len(parts), dtype=np.int64)
        if H > 31:
            raise ValueError("more than 31 haplotypes cannot be stored in the .bin bitmask (bin_utils.py:208-210)")
        self.n_loci, self.n_haplotypes = T, H


def record_valid(flag, tid, next_tid, next_pos):
    """Vectorised record filter (``bam_utils.py:264-270``)."""
    flag = flag.astype(np.int64)
    paired = (flag & 0x1) != 0
    bad = ((flag & 0x80) != 0) | ((flag & 0x2) == 0) | (tid != next_tid) | (next_pos < 0)
    return ((flag & 0x4) == 0) & ~(paired & bad)


def trim_name(q):
    """Cut at the first space only if its index is > 0 (``bam_utils.py:292-294``)."""
    i = q.find(' ')
    return q[:i] if i > 0 else q


class TupleEncoder(object):
    """Streams decoded BAM records into device tuples; carries the open read across batches."""

    def __init__(self, maps, trim=True):
        self.maps = maps
        self.cur = 0xFFFFFFFF            # id of the latest read started (none yet)
        self.last_name = None            # its (trimmed) name
        self.trim = trim
        self.names_of_reads = None       # optional: first name of every read (multisample needs the cell)

    def encode(self, qnames, flag, tid, pos, next_tid, next_pos):
        n = len(qnames)
        flag = np.asarray(flag)
        tid = np.asarray(tid, dtype=np.int64)
        valid = record_valid(flag, tid, np.asarray(next_tid, dtype=np.int64), np.asarray(next_pos, dtype=np.int64))
        vi = np.nonzero(valid)[0]
        # a read starts at a valid record whose (trimmed) name differs from the previous valid record's (bam_utils.py:289-320).
        # No Python loop over records: the names become one fixed-width code-point matrix, each row is blanked from its cut
        # (first space at an index > 0, bam_utils.py:292-294) on, and consecutive rows are compared.
        head = np.zeros(len(vi), dtype=bool)
        if len(vi):
            names = np.asarray(qnames, dtype=str)[vi] if not isinstance(qnames, np.ndarray) else qnames[vi].astype(str)
            width = max(names.dtype.itemsize // 4, 1)
            lens = np.char.str_len(names)
            cut = lens
            if self.trim:
                sp = np.char.find(names, ' ')
                cut = np.where(sp > 0, sp, lens)
            code = np.ascontiguousarray(names).view(np.uint32).reshape(len(names), width)
            code = np.where(np.arange(width)[None, :] < cut[:, None], code, 0)
            head[1:] = (cut[1:] != cut[:-1]) | (code[1:] != code[:-1]).any(axis=1)
            first = str(names[0][:cut[0]])
            head[0] = first != self.last_name
            self.last_name = str(names[-1][:cut[-1]])
            if self.names_of_reads is not None:
                self.names_of_reads.extend(str(x) for x in np.asarray(qnames, dtype=str)[vi[head]])
        starts = np.zeros(n, dtype=np.int64)
        starts[vi[head]] = 1
        rid = (np.cumsum(starts) + (np.int64(self.cur) if self.cur != 0xFFFFFFFF else -1))
        if n:
            last = int(rid[-1])
            self.cur = last if last >= 0 else 0xFFFFFFFF
        safe_tid = np.where(valid, tid, 0)
        locus = self.maps.tid2locus[safe_tid]
        hap = self.maps.tid2hap[safe_tid]
        hostbits = np.where(tid != np.asarray(next_tid, dtype=np.int64), FLAG_MATE_OTHER_REF, 0) | \
            np.where(np.asarray(next_pos, dtype=np.int64) < 0, FLAG_NEXT_POS_NEG, 0)
        hapflag = (flag.astype(np.int64) & 0xFFF) | hostbits | (hap.astype(np.int64) << HAP_SHIFT)
        return dict(read_id=(rid & 0xFFFFFFFF).astype(np.uint32), locus=locus.astype(np.uint32),
                    hapflag=hapflag.astype(np.uint32), pos=np.asarray(pos, dtype=np.int32), n_valid=int(valid.sum()))

    def encode_decoded(self, flag, tid, pos, next_tid, next_pos, valid, head):
        """The same tuples from what ``bamdec.NativeBamReader.read_decoded`` returns: the filter verdict and the read heads come
        from the decoder (which compares the names where they are, in the inflated BAM stream), the rest is array arithmetic."""
        flag = np.asarray(flag)
        tid = np.asarray(tid, dtype=np.int64)
        valid = np.asarray(valid).astype(bool)
        n = len(flag)
        rid = np.cumsum(np.asarray(head, dtype=np.int64)) + (np.int64(self.cur) if self.cur != 0xFFFFFFFF else -1)
        if n:
            last = int(rid[-1])
            self.cur = last if last >= 0 else 0xFFFFFFFF
        safe_tid = np.where(valid, tid, 0)
        locus = self.maps.tid2locus[safe_tid]
        hap = self.maps.tid2hap[safe_tid]
        hostbits = np.where(tid != np.asarray(next_tid, dtype=np.int64), FLAG_MATE_OTHER_REF, 0) | \
            np.where(np.asarray(next_pos, dtype=np.int64) < 0, FLAG_NEXT_POS_NEG, 0)
        hapflag = (flag.astype(np.int64) & 0xFFF) | hostbits | (hap.astype(np.int64) << HAP_SHIFT)
        return dict(read_id=(rid & 0xFFFFFFFF).astype(np.uint32), locus=locus.astype(np.uint32),
                    hapflag=hapflag.astype(np.uint32), pos=np.asarray(pos, dtype=np.int32).copy(), n_valid=int(valid.sum()))
