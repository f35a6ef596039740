# -*- coding: utf-8 -*-
"""Deterministic synthetic alignment streams for tests and ``bench.py``.

Every value is a pure function of ``(seed, global read index, k)`` through a
counter-based splitmix64, so any contiguous read shard regenerates exactly its
slice of the whole stream (multi-GPU shards, CPU-baseline samples, fixtures).
The same code runs on numpy (CPU) and on torch tensors (GPU): all arithmetic is
wrap-around int64 with explicit logical shifts, so the two agree bit for bit.

Workload model (BASELINE.json configs):

* targets are ``T`` loci x ``H`` haplotypes, BAM ``tid = locus*H + hap`` and
  reference names ``ENSMUST%011d_%c``; ``length(locus) = 500 + locus*7919 % 4500``;
* a read picks a log-uniform (Zipf-like, s~1) base locus, ``n_loci`` in
  ``{1,1,2,3,5}`` consecutive loci and one of ``n_variants`` presence variants;
  haplotype ``h`` of locus ``l`` is present in variant ``v`` w.p. 0.85 (hashed,
  at least one forced), giving ~16 valid alignments per read at ``H = 8``;
* 2 % of reads repeat their first alignment (the reference collapses duplicate
  (read, target) pairs, ``bam_utils.py:322-325``);
* single-end: 0-2 unmapped records (flag 4) are interleaved into a read
  (~5 % of all records) and 4 % of reads are entirely unmapped;
* paired-end: every alignment is a read1 record (0x43/0x53) followed by its
  read2 mate (0x83/0x93); 3 % of the reads are not properly paired and 1 % have
  the mate on another reference (all their alignments, so the read vanishes),
  and 0.2 % + 0.2 % of single alignments have the same problems -- all dropped
  by the reference's filter (``bam_utils.py:268-270``);
* the alignments of a read are rotated by a per-read offset so they do not
  arrive sorted by target.

Outputs are the *raw BAM fields* per record (what pysam would hand the
reference) plus the device tuple form of ``include/ecb.h``.
"""
from __future__ import annotations

import numpy as np

SEED = 20260101
GEN_VERSION = 2  # bump when the stream changes: committed fixtures depend on it

_NLOCI_CHOICES = (1, 1, 2, 3, 5)
_MAXL = 5
_MAXP = 3   # further loci a read may hit anywhere among the targets ("paralogs", SynthSpec.paralog_pct)
_M64 = (1 << 64) - 1


def _i64(c):
    """Python int (taken mod 2**64) as a signed 64-bit Python int."""
    c &= _M64
    return c - (1 << 64) if c >= (1 << 63) else c


_C_GAMMA = _i64(0x9E3779B97F4A7C15)
_C_M1 = _i64(0xBF58476D1CE4E5B9)
_C_M2 = _i64(0x94D049BB133111EB)


class _NP(object):
    """numpy backend (int64 arrays, wrap-around arithmetic)."""
    name = "numpy"

    @staticmethod
    def arange(a, b):
        return np.arange(a, b, dtype=np.int64)

    @staticmethod
    def full(n, v):
        return np.full(n, v, dtype=np.int64)

    @staticmethod
    def where(c, a, b):
        return np.where(c, a, b)

    @staticmethod
    def cumsum(x):
        return np.cumsum(x, dtype=np.int64)

    @staticmethod
    def repeat(x, counts, total):
        return np.repeat(x, counts)

    @staticmethod
    def cat(xs):
        return np.concatenate(xs)

    @staticmethod
    def take(lut, idx):
        return lut[idx]

    @staticmethod
    def lut(values):
        return np.asarray(values, dtype=np.int64)

    @staticmethod
    def to(x, dt):
        return x.astype(dt)

    i32, u32, u16, i64 = np.int32, np.uint32, np.uint16, np.int64


class _Torch(object):
    """torch backend (int64 tensors on ``device``)."""
    name = "torch"

    def __init__(self, device):
        import torch
        self.t = torch
        self.device = device
        self.i32, self.i64 = torch.int32, torch.int64
        self.u32 = torch.int32   # reinterpreted by the consumer (same bits)
        self.u16 = torch.int16

    def arange(self, a, b):
        return self.t.arange(a, b, dtype=self.t.int64, device=self.device)

    def full(self, n, v):
        return self.t.full((n,), v, dtype=self.t.int64, device=self.device)

    def where(self, c, a, b):
        if not self.t.is_tensor(a):
            a = self.t.full_like(b if self.t.is_tensor(b) else c, a, dtype=self.t.int64)
        if not self.t.is_tensor(b):
            b = self.t.full_like(a, b)
        return self.t.where(c, a, b)

    def cumsum(self, x):
        return self.t.cumsum(x, 0)

    def repeat(self, x, counts, total):
        return self.t.repeat_interleave(x, counts, output_size=int(total))

    def cat(self, xs):
        return self.t.cat(xs)

    def take(self, lut, idx):
        return lut[idx]

    def lut(self, values):
        return self.t.tensor(values, dtype=self.t.int64, device=self.device)

    def to(self, x, dt):
        return x.to(dt)


def _lsr(x, s):
    """Logical right shift of int64 lanes."""
    return (x >> s) & ((1 << (64 - s)) - 1)


def _mix(x):
    """splitmix64 finaliser on wrap-around int64 lanes."""
    with np.errstate(over="ignore"):
        z = x + _C_GAMMA
        z = (z ^ _lsr(z, 30)) * _C_M1
        z = (z ^ _lsr(z, 27)) * _C_M2
        return z ^ _lsr(z, 31)


def _rnd(seed, a, k):
    """Counter-based random int64 lanes (use ``_pos`` before a modulo)."""
    with np.errstate(over="ignore"):
        return _mix(_mix(a * _i64(0xD1342543DE82EF95) + _i64(seed * 0x2545F4914F6CDD1D + k)) + k)


def _pos(x):
    return _lsr(x, 1)  # non-negative 63-bit value


def _kth_set_bit_lut():
    lut = []
    for byte in range(256):
        for k in range(8):
            bits = [b for b in range(8) if byte >> b & 1]
            lut.append(bits[k] if k < len(bits) else 0)
    return lut


def _popcount8_lut():
    return [bin(b).count("1") for b in range(256)]


class SynthSpec(object):
    """Parameters of one synthetic workload."""

    def __init__(self, n_reads, n_loci, n_haps, paired=False, seed=SEED, n_variants=4,
                 unmapped_read_pct=4, dup_pct=2, locus_stride=1, paralog_pct=0):
        if not 1 <= n_haps <= 8:
            raise ValueError("synthetic generator supports 1..8 haplotypes")
        self.n_reads, self.n_loci, self.n_haps = int(n_reads), int(n_loci), int(n_haps)
        self.paired, self.seed, self.n_variants = bool(paired), int(seed), int(n_variants)
        self.unmapped_read_pct, self.dup_pct = int(unmapped_read_pct), int(dup_pct)
        self.locus_stride = int(locus_stride)       # a read's loci are base, base + stride, ... (1: consecutive target ids, the default and what the fixtures hold)
        # this percentage of the reads hit 1 - 3 further loci besides their cluster: the cluster's "paralogs", three loci drawn uniformly
        # over all targets per base locus (multi-mappers at unrelated target ids; 0: none, the default and what the fixtures hold --
        # the stream is then unchanged)
        self.paralog_pct = int(paralog_pct)

    # header --------------------------------------------------------------
    def hap_names(self):
        return [chr(ord("A") + h) for h in range(self.n_haps)]

    def locus_length(self, locus):
        return 500 + (locus * 7919) % 4500

    def references(self):
        """``[(name, length)]`` in tid order (``tid = locus*H + hap``)."""
        haps = self.hap_names()
        return [("ENSMUST%011d_%s" % (l, h), self.locus_length(l))
                for l in range(self.n_loci) for h in haps]

    def read_name(self, r):
        return "r%010d" % r


def _per_read(spec, be, r):
    """Per-read quantities for global read indices ``r`` (int64 lanes)."""
    T, H, seed = spec.n_loci, spec.n_haps, spec.seed
    nbits = max(1, (T - 1).bit_length())
    o = _pos(_rnd(seed, r, 0)) % nbits
    base = (((o * 0 + 1) << o) + _pos(_rnd(seed, r, 1)) % ((o * 0 + 1) << o) - 1) % T
    nl = be.take(be.lut(_NLOCI_CHOICES), _pos(_rnd(seed, r, 2)) % 5)
    if T < _MAXL:
        nl = be.where(nl > T, T, nl)
    v = _pos(_rnd(seed, r, 3)) % spec.n_variants
    pc = be.lut(_popcount8_lut())
    masks, cnts, locs = [], [], []
    nslots = _MAXL + (_MAXP if spec.paralog_pct else 0)
    if spec.paralog_pct:
        n_par = be.where(_pos(_rnd(seed, r, 20)) % 100 < spec.paralog_pct, 1 + _pos(_rnd(seed, r, 21)) % _MAXP, 0)
    for j in range(nslots):
        if j < _MAXL:
            loc = (base + j * spec.locus_stride) % T
            here = nl > j
        else:
            loc = _pos(_rnd(seed ^ 0x9A7A, base * 8 + j, 22)) % T     # (a property of the cluster, not of the read: paralogs are fixed relations, and the ECs stay few)
            here = n_par > (j - _MAXL)
        m = loc * 0
        for h in range(H):
            hv = _pos(_rnd(seed ^ 0x5EED, loc * 64 + v * 8 + h, 7)) % 100
            m = m | be.where(hv < 85, 1 << h, 0)
        forced = (m * 0 + 1) << (_pos(_rnd(seed ^ 0x5EED, loc * 64 + v * 8, 9)) % H)
        m = be.where(m == 0, forced, m)
        m = be.where(here, m, 0)
        masks.append(m)
        cnts.append(be.take(pc, m))
        locs.append(loc)
    n_al = cnts[0] + cnts[1] + cnts[2] + cnts[3] + cnts[4]
    for j in range(_MAXL, nslots):
        n_al = n_al + cnts[j]
    dup = (_pos(_rnd(seed, r, 4)) % 100 < spec.dup_pct)
    n_al_d = n_al + be.where(dup, 1, 0)           # alignments incl. the duplicate
    rot = _pos(_rnd(seed, r, 5)) % n_al
    if spec.paired:
        unm_read = n_al < 0                        # none
        n_unm = n_al * 0
        n_rec = 2 * n_al_d
    else:
        unm_read = (_pos(_rnd(seed, r, 6)) % 100 < spec.unmapped_read_pct)
        e = _pos(_rnd(seed, r, 8)) % 100
        n_unm = be.where(e < 15, 2, be.where(e < 50, 1, 0))
        n_rec = be.where(unm_read, 1, n_al_d + n_unm)
    upos = _pos(_rnd(seed, r, 10)) % n_al_d        # unmapped records sit before alignment `upos`
    return dict(base=base, nl=nl, masks=masks, cnts=cnts, locs=locs, n_al=n_al, n_al_d=n_al_d, dup=dup,
                rot=rot, unm_read=unm_read, n_unm=n_unm, n_rec=n_rec, upos=upos)


def count_records(spec, r0, r1, device=None):
    """Number of BAM records of reads ``[r0, r1)``."""
    be = _Torch(device) if device is not None else _NP()
    tot = 0
    step = 1 << 22
    for a in range(r0, r1, step):
        pr = _per_read(spec, be, be.arange(a, min(a + step, r1)))
        tot += int(pr["n_rec"].sum())
    return tot


def generate(spec, r0, r1, device=None, want_raw=False, read_id_base=0):
    """Records of reads ``[r0, r1)``.

    Returns a dict of per-record columns:

    * device tuple form (``include/ecb.h``): ``read_id`` (run counter over valid
      records, forward-filled; numbering starts at ``read_id_base``), ``locus``,
      ``hapflag`` (= ``flag | hap << 16``; flag bit 12 = mate on another reference,
      bit 13 = ``next_pos < 0``), ``pos``;
    * with ``want_raw``: ``read`` (global read index, names the qname), ``flag``,
      ``tid``, ``next_tid``, ``next_pos`` -- the raw BAM fields.

    numpy arrays, or torch tensors on ``device`` (``read_id``/``locus``/``hapflag``
    are int32 tensors holding the uint32 bit patterns).
    """
    be = _Torch(device) if device is not None else _NP()
    T, H, seed = spec.n_loci, spec.n_haps, spec.seed
    r = be.arange(r0, r1)
    pr = _per_read(spec, be, r)
    n_rec = pr["n_rec"]
    total = int(n_rec.sum())
    ends = be.cumsum(n_rec)
    starts = ends - n_rec
    ridx = be.repeat(be.arange(0, r1 - r0), n_rec, total)       # local read index per record
    w = be.arange(0, total) - be.take(starts, ridx)             # index within the read
    g = lambda x: be.take(x, ridx)                              # noqa: E731
    rr = g(r)
    n_al, n_al_d, rot = g(pr["n_al"]), g(pr["n_al_d"]), g(pr["rot"])
    if spec.paired:
        a_d = w >> 1                                            # alignment slot (incl. dup)
        mate2 = (w & 1) == 1
        is_unm = w < 0
    else:
        n_unm, upos, unm_read = g(pr["n_unm"]), g(pr["upos"]), g(pr["unm_read"])
        is_unm = unm_read | ((w >= upos) & (w < upos + n_unm))
        a_d = be.where(w >= upos + n_unm, w - n_unm, w)
        a_d = be.where(is_unm, 0, a_d)
        mate2 = w < 0
    # alignment slot -> ordinal among the read's set (locus, hap) bits
    # (the extra slot of a "duplicate" read repeats the read's first alignment)
    a = be.where(a_d >= n_al, rot, (a_d + rot) % n_al)
    # locate ordinal `a`: which locus byte, then k-th set bit of that byte
    kth = be.lut(_kth_set_bit_lut())
    j = a * 0
    rem = a
    done = a < 0
    mask_sel = a * 0
    loc_sel = a * 0
    for jj in range(len(pr["cnts"])):
        c = g(pr["cnts"][jj])
        hit = (~done) & (rem < c)
        j = be.where(hit, jj, j)
        mask_sel = be.where(hit, g(pr["masks"][jj]), mask_sel)
        if spec.paralog_pct:
            loc_sel = be.where(hit, g(pr["locs"][jj]), loc_sel)
        rem = be.where(done | hit, rem, rem - c)
        done = done | hit
    hap = be.take(kth, mask_sel * 8 + rem)
    locus = loc_sel if spec.paralog_pct else (g(pr["base"]) + j * spec.locus_stride) % T
    tid = locus * H + hap
    length = 500 + (locus * 7919) % 4500
    pos = _pos(_rnd(seed, rr * 64 + a, 11)) % length
    rev = _pos(_rnd(seed, rr * 64 + a, 12)) & 1
    if spec.paired:
        # pairing problems are mostly a property of the fragment (all its alignments), rarely of one alignment
        qr = _pos(_rnd(seed, rr, 14)) % 100
        q = _pos(_rnd(seed, rr * 64 + a, 13)) % 1000
        improper = (qr < 3) | (q < 2)
        other = ((qr >= 3) & (qr < 4)) | ((q >= 2) & (q < 4))
        flag = be.where(mate2, 0x81, 0x41) + be.where(improper, 0, 0x2) + \
            be.where(rev == 1, be.where(mate2, 0x20, 0x10), be.where(mate2, 0x10, 0x20))
        next_tid = be.where(other, (tid + 1) % (T * H), tid)
        next_pos = (pos + 150) % length
        hostbits = be.where(next_tid != tid, 1 << 12, 0)
    else:
        flag = be.where(is_unm, 0x4, be.where(rev == 1, 0x10, 0))
        tid = be.where(is_unm, -1, tid)
        locus = be.where(is_unm, 0, locus)
        hap = be.where(is_unm, 0, hap)
        pos = be.where(is_unm, -1, pos)
        next_tid = tid * 0 - 1
        next_pos = tid * 0 - 1
        hostbits = tid * 0
    valid = ((flag & 0x4) == 0) & (((flag & 0x1) == 0) |
                                   (((flag & 0x80) == 0) & ((flag & 0x2) != 0) &
                                    (next_tid == tid) & (next_pos >= 0)))
    # run counter over valid records (every synthetic read has a distinct name), forward-filled
    nv = _seg_sum(be, valid, ridx, r1 - r0)                     # valid records per read
    first_valid = valid & (be.cumsum(be.where(valid, 1, 0)) - g(be.cumsum(nv) - nv) == 1)
    read_id = be.cumsum(be.where(first_valid, 1, 0)) - 1 + read_id_base
    hapflag = flag | hostbits | (hap << 16)
    out = dict(read_id=be.to(read_id, be.u32), locus=be.to(locus, be.u32),
               hapflag=be.to(hapflag, be.u32), pos=be.to(pos, be.i32),
               n_records=total, n_valid=int(valid.sum()),
               n_reads=int(first_valid.sum()))
    if want_raw:
        out.update(read=rr, flag=be.to(flag, be.u16), tid=be.to(tid, be.i32),
                   next_tid=be.to(next_tid, be.i32), next_pos=be.to(next_pos, be.i32))
    return out


def _seg_sum(be, flags, seg, nseg):
    """Per-segment count of set ``flags`` (segments = ``seg`` ids, sorted)."""
    if be.name == "numpy":
        return np.bincount(seg[flags], minlength=nseg).astype(np.int64)
    t = be.t
    return t.zeros(nseg, dtype=t.int64, device=be.device).index_add_(
        0, seg, flags.to(t.int64))


def raw_records(spec, r0, r1):
    """Iterator of ``(qname, flag, tid, pos, next_tid, next_pos)`` for ``bamio.write_bam``."""
    g = generate(spec, r0, r1, want_raw=True)
    for i in range(g["n_records"]):
        yield (spec.read_name(int(g["read"][i])), int(g["flag"][i]), int(g["tid"][i]),
               int(g["pos"][i]), int(g["next_tid"][i]), int(g["next_pos"][i]))
