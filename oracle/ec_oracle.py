# -*- coding: utf-8 -*-
"""ORACLE -- test infrastructure, not product code.

CPU restatement of alntools' bam2ec / bam2emase hot path (reference v0.1.1,
paths relative to /root/reference/alntools).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product (``alntools_amd``) never does.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function
here against fixtures produced by running the reference itself
(``tests/golden/make_golden.py``): byte-identical ``.bin`` and range files for the
edge-case, config-1, 8-haplotype, paired-end and multisample cases.  Two things
are NOT pinned and cannot be here: decode-level agreement with the real
pysam/htslib (not installed; fixtures are pinned at the decoded-field level on
BAMs this repo writes) and the EMASE ``.h5`` container (PyTables absent).

The algorithm is restated the slow, literal way on purpose: per-alignment loop,
string keys, ordered dicts, scipy for the sparse canonicalisation the reference
delegates to scipy (bam_utils.py:837-847, bin_utils.py:208-211).
"""
from __future__ import annotations

from collections import OrderedDict
from struct import pack, unpack_from

import numpy as np
from scipy.sparse import coo_matrix, csc_matrix, csr_matrix

RANGE_UNSEEN = (100000000000, -1)          # bam_utils.py:283


# --------------------------------------------------------------------------- #
# a1  header -> index maps                                   bam_utils.py:561-633
# --------------------------------------------------------------------------- #
def parse_targets(text):
    """utils.py:161-178 -- first whitespace token of each line not starting with '#'."""
    targets = OrderedDict()
    for line in text.splitlines(True):
        if line and line[0] == "#":
            continue
        targets[line.strip().split()[0]] = len(targets)
    return targets


def split_reference_name(name):
    """bam_utils.py:584-591 -- split at the LAST '_' if its index is > 0."""
    i = name.rfind("_")
    if i > 0:
        return name[:i], name[i + 1:]
    return name, ""


def header_maps(references, lengths, targets_txt=None):
    main_targets = OrderedDict()
    if targets_txt:
        main_targets = parse_targets(targets_txt)                    # :571-579
    tid_of = {}
    for tid, name in enumerate(references):
        tid_of.setdefault(name, tid)                                 # get_tid: first match
    tid_to_target = {}
    haps = set()
    for tid, name in enumerate(references):
        target, hap = split_reference_name(name)
        tid_to_target[str(tid_of[name])] = target                    # :583,593
        haps.add(hap)
        if target not in main_targets:                               # :596-598
            main_targets[target] = len(main_targets)
    haplotypes = sorted(haps)                                        # :602
    hap_idx = {h: i for i, h in enumerate(haplotypes)}
    lens = np.zeros((len(main_targets), len(haplotypes)), dtype=np.int32)   # :605
    for tid, name in enumerate(references):                          # :615-633
        target, hap = split_reference_name(name)
        lens[main_targets[target], hap_idx[hap]] = lengths[tid]
    return dict(main_targets=main_targets, haplotypes=haplotypes, tid_to_target=tid_to_target,
                lengths=lens, tid_of=tid_of, references=list(references))


def gettid(maps, name):
    return maps["tid_of"].get(name, -1)


def transcript_name(main_target, hap):
    """bam_utils.py:749-752, 801-806."""
    return main_target if len(hap) == 0 else "{}_{}".format(main_target, hap)


# --------------------------------------------------------------------------- #
# a2  record filter                                          bam_utils.py:264-270
# --------------------------------------------------------------------------- #
def record_is_valid(flag, tid, next_tid, next_pos):
    if flag & 0x4:
        return False
    if flag & 0x1:
        if (flag & 0x80) or not (flag & 0x2) or tid != next_tid or next_pos < 0:
            return False
    return True


def trim_name(q):
    """bam_utils.py:292-294 -- cut at the first space only if its index is > 0."""
    i = q.find(" ")
    return q[:i] if i > 0 else q


# --------------------------------------------------------------------------- #
# a3-a5  per-alignment scan of one chunk                     bam_utils.py:198-363
# --------------------------------------------------------------------------- #
def scan(records, track_ranges=False):
    """``records``: iterable of ``(qname, flag, tid, pos, next_tid, next_pos)``.

    -> dict(ec=OrderedDict key->count, all, valid, unique_reads, ranges).
    Raises ValueError where the reference's worker dies (no valid alignment in the
    chunk: ``','.join([None])``, bam_utils.py:336-339 -- SURVEY 8a-Q8).
    """
    ec = OrderedDict()
    unique_reads = {}
    ranges = {}
    n_all = n_valid = 0
    query_name = None
    reference_id = None
    reference_ids = []
    for (qname, flag, tid, pos, next_tid, next_pos) in records:
        n_all += 1
        if not record_is_valid(flag, tid, next_tid, next_pos):
            continue
        n_valid += 1
        reference_id = str(tid)
        if track_ranges:                                             # :282-286
            lo, hi = ranges.get(reference_id, RANGE_UNSEEN)
            ranges[reference_id] = (min(lo, pos), max(hi, pos))
        if query_name is None:
            query_name = trim_name(qname)
        unique_reads[query_name] = unique_reads.get(query_name, 0) + 1    # :296-299 (before the switch)
        if query_name != trim_name(qname):                           # :306-320
            key = ",".join(sorted(reference_ids))
            ec[key] = ec.get(key, 0) + 1
            query_name = trim_name(qname)
            reference_ids = [reference_id]
        elif reference_id not in reference_ids:                      # :322-325
            reference_ids.append(reference_id)
    if reference_id is None:
        raise ValueError("no valid alignments (the reference fails here: bam_utils.py:336-339)")
    if reference_id not in reference_ids:                            # :336-337
        reference_ids.append(reference_id)
    key = ",".join(sorted(reference_ids))                            # :339-344
    ec[key] = ec.get(key, 0) + 1
    return dict(ec=ec, all=n_all, valid=n_valid, unique_reads=len(unique_reads), ranges=ranges)


def merge_scans(results):
    """bam_utils.py:680-724 -- ordered union; EC rank = global first appearance."""
    final = OrderedDict()
    ranges = {}
    n_all = n_valid = 0
    for r in results:
        for k, v in r["ec"].items():
            final[k] = final.get(k, 0) + v
        n_all += r["all"]
        n_valid += r["valid"]
        for k, (lo, hi) in r["ranges"].items():
            if k in ranges:
                ranges[k] = (min(lo, ranges[k][0]), max(hi, ranges[k][1]))
            else:
                ranges[k] = (lo, hi)
    return dict(ec=final, all=n_all, valid=n_valid, ranges=ranges)


# --------------------------------------------------------------------------- #
# f-3  range file                                            bam_utils.py:735-766
# --------------------------------------------------------------------------- #
def range_text(maps, ranges):
    out = ["#\t" + "\t".join(maps["haplotypes"]) + "\n"]
    for main_target in maps["main_targets"]:
        vals = []
        for hap in maps["haplotypes"]:
            tid = str(gettid(maps, transcript_name(main_target, hap)))
            mm = ranges.get(tid)
            if mm is None or mm == RANGE_UNSEEN:
                vals.append("0")
            else:
                vals.append(str(mm[1] - mm[0] + 1))
        out.append(main_target + "\t" + "\t".join(vals) + "\n")
    return "".join(out)


# --------------------------------------------------------------------------- #
# a7-a9  incidence + count matrices                          bam_utils.py:768-847
# --------------------------------------------------------------------------- #
def build_incidence(maps, ec_keys):
    """Per-haplotype CSC matrices (E x T), as ``apm.finalize()`` leaves them."""
    mt, haps = maps["main_targets"], maps["haplotypes"]
    rows = [[] for _ in haps]
    cols = [[] for _ in haps]
    for e, key in enumerate(ec_keys):
        tids = key.split(",")
        for main_target in set(maps["tid_to_target"][t] for t in tids):      # :792-797
            for i, hap in enumerate(haps):
                if str(gettid(maps, transcript_name(main_target, hap))) in tids:   # :809-811
                    rows[i].append(e)
                    cols[i].append(mt[main_target])
    shape = (len(ec_keys), len(mt))
    return [coo_matrix((np.ones(len(rows[i])), (rows[i], cols[i])), shape=shape).tocsc()
            for i in range(len(haps))]


def combined_csr(data):
    """bin_utils.py:208-211 -- A = sum_h 2^h * data[h], as CSR."""
    a = data[0]
    for h in range(1, len(data)):
        a = a + ((2 ** h) * data[h])
    return a.tocsr()


# --------------------------------------------------------------------------- #
# a10 / f-1  .bin (EC format 2)                              bin_utils.py:105-277, 32-102
# --------------------------------------------------------------------------- #
def _ints(a):
    return np.asarray(a).astype(int).astype("<i4").tobytes()


def _name(s):
    return pack("<i", len(s)) + pack("<{}s".format(len(s)), s.encode("utf-8"))


def ecsave2_bytes(hname, lname, lengths, sname, data, count):
    """``data``: per-haplotype sparse matrices; ``count``: scipy CSC (E x S)."""
    out = [pack("<i", 2), pack("<i", len(hname))]
    out += [_name(h) for h in hname]
    out.append(pack("<i", len(lname)))
    lens = np.asarray(lengths).astype(int)
    for t, name in enumerate(lname):
        out.append(_name(name))
        out.append(_ints(lens[t, :len(hname)]))
    out.append(pack("<i", len(sname)))
    out += [_name(s) for s in sname]
    a = combined_csr(data)
    out += [pack("<i", len(a.indptr)), pack("<i", a.nnz), _ints(a.indptr), _ints(a.indices), _ints(a.data)]
    n = count
    out += [pack("<i", len(n.indptr)), pack("<i", n.nnz), _ints(n.indptr), _ints(n.indices), _ints(n.data)]
    return b"".join(out)


def ecload_bytes(b):
    """Walk a format-2 ``.bin``; -> dict of names, lengths and the raw A (CSR) / N (CSC) arrays."""
    o = [0]

    def i32(n=1):
        v = np.frombuffer(b, dtype="<i4", count=n, offset=o[0])
        o[0] += 4 * n
        return v

    def name():
        n = int(i32()[0])
        s = unpack_from("<{}s".format(n), b, o[0])[0].decode("utf-8")
        o[0] += n
        return s

    if int(i32()[0]) != 2:
        raise TypeError("only EC format 2 is supported (bin_utils.py:98-102)")
    H = int(i32()[0])
    hname = [name() for _ in range(H)]
    T = int(i32()[0])
    lname, lens = [], np.zeros((T, H), dtype=np.int64)
    for t in range(T):
        lname.append(name())
        lens[t] = i32(H)
    S = int(i32()[0])
    sname = [name() for _ in range(S)]
    na, nnz = int(i32()[0]), int(i32()[0])
    A = (i32(na).copy(), i32(nnz).copy(), i32(nnz).copy())
    nn, nnzn = int(i32()[0]), int(i32()[0])
    N = (i32(nn).copy(), i32(nnzn).copy(), i32(nnzn).copy())
    if o[0] != len(b):
        raise ValueError("trailing bytes in .bin")
    return dict(hname=hname, lname=lname, lengths=lens, sname=sname,
                indptrA=A[0], indicesA=A[1], dataA=A[2], indptrN=N[0], indicesN=N[1], dataN=N[2])


# --------------------------------------------------------------------------- #
# single-sample driver                                       bam_utils.py:512-876
# --------------------------------------------------------------------------- #
def convert_records(references, lengths, records, sample, targets_txt=None, want_range=False):
    """-> dict(bin=bytes, range=str|None, counters, ec=OrderedDict)."""
    maps = header_maps(references, lengths, targets_txt)
    res = merge_scans([scan(records, track_ranges=want_range)])
    keys = list(res["ec"].keys())
    data = build_incidence(maps, keys)
    count = csc_matrix(np.matrix(list(res["ec"].values())).T)        # :845
    b = ecsave2_bytes(maps["haplotypes"], list(maps["main_targets"].keys()), maps["lengths"],
                      [sample], data, count)
    return dict(bin=b, range=range_text(maps, res["ranges"]) if want_range else None,
                counters={"all": res["all"], "valid": res["valid"], "ecs": len(keys)},
                ec=res["ec"], maps=maps)


# --------------------------------------------------------------------------- #
# a11-a12  multisample                        bam_utils_multisample.py:175-321, 503-820
# --------------------------------------------------------------------------- #
def scan_multisample(records, track_ranges=False):
    """One BAM file.  Quirks kept: the run being closed supplies the
    cell (``split('|||')[14]`` of the TRACKED name, :270-280); after a switch the tracked
    name is the UNTRIMMED query name (:292); the last read of the file is never counted."""
    ec = OrderedDict()
    ranges = {}
    read_ids = {}
    n_all = n_valid = 0
    query_name = None
    reference_ids = []
    for (qname, flag, tid, pos, next_tid, next_pos) in records:
        n_all += 1
        if not record_is_valid(flag, tid, next_tid, next_pos):
            continue
        n_valid += 1
        reference_id = str(tid)
        if track_ranges:
            lo, hi = ranges.get(reference_id, RANGE_UNSEEN)
            ranges[reference_id] = (min(lo, pos), max(hi, pos))
        if query_name is None:
            query_name = trim_name(qname)
        read_ids[query_name] = read_ids.get(query_name, 0) + 1
        cell = query_name.split("|||")[14]
        if query_name != trim_name(qname):
            key = ",".join(sorted(reference_ids))
            d = ec.setdefault(key, OrderedDict())
            d[cell] = d.get(cell, 0) + 1
            query_name = qname                                       # untrimmed (:292)
            reference_ids = [reference_id]
        elif reference_id not in reference_ids:
            reference_ids.append(reference_id)
    return dict(ec=ec, all=n_all, valid=n_valid, ranges=ranges, reads=len(read_ids))


def convert_multisample(references, lengths, files, minimum_count, targets_txt=None, want_range=False):
    """``files``: list of record lists, in the order the reference's glob returned them."""
    maps = header_maps(references, lengths, targets_txt)
    final = OrderedDict()
    cr_totals = OrderedDict()
    ranges = {}
    n_all = n_valid = 0
    for recs in files:                                               # :503-576
        r = scan_multisample(recs, want_range)
        for key, cells in r["ec"].items():
            for cell, c in cells.items():
                cr_totals[cell] = cr_totals.get(cell, 0) + c
                d = final.setdefault(key, OrderedDict())
                d[cell] = d.get(cell, 0) + c
        n_all += r["all"]
        n_valid += r["valid"]
        for k, (lo, hi) in r["ranges"].items():
            ranges[k] = (min(lo, ranges[k][0]), max(hi, ranges[k][1])) if k in ranges else (lo, hi)
    if minimum_count <= 0:                                           # :596-597
        minimum_count = 1
    crs = OrderedDict()
    for cell, tot in cr_totals.items():                              # :605-608
        if tot >= minimum_count:
            crs[cell] = len(crs)
    kept = OrderedDict()
    for key, cells in final.items():                                 # :616-632
        sub = OrderedDict((c, n) for c, n in cells.items() if c in crs)
        if sub:
            kept[key] = sub
    keys = list(kept.keys())
    data = build_incidence(maps, keys)
    indptr, indices, vals = [0], [], []
    for key in keys:                                                 # :738-747
        for cell in sorted(kept[key].keys(), key=lambda c: crs[c]):
            indices.append(crs[cell])
            vals.append(kept[key][cell])
        indptr.append(len(indices))
    npa = csr_matrix((np.array(vals, dtype=np.int32), np.array(indices, dtype=np.int32),
                      np.array(indptr, dtype=np.int32)), shape=(len(keys), len(crs)))
    b = ecsave2_bytes(maps["haplotypes"], list(maps["main_targets"].keys()), maps["lengths"],
                      list(crs.keys()), data, npa.tocsc())
    return dict(bin=b, range=range_text(maps, ranges) if want_range else None,
                counters={"all": n_all, "valid": n_valid, "ecs": len(keys), "cells": len(crs),
                          "ecs_before": len(final), "cells_before": len(cr_totals)},
                samples=list(crs.keys()))


# --------------------------------------------------------------------------- #
# utils                                                              utils.py:67-133
# --------------------------------------------------------------------------- #
def partition(lst, n):
    q, r = divmod(len(lst), n)
    cuts = [q * i + min(i, r) for i in range(n + 1)]
    out = []
    for i in range(n):
        part = lst[cuts[i]:cuts[i + 1]]
        if not part:
            break
        out.append(part)
    return out


def list_to_int(lst):
    c = 0
    for i, on in enumerate(lst):
        if on == 1:
            c |= 1 << i
    return c


def int_to_list(c, size):
    return [1 if c & (1 << i) else 0 for i in range(size)]


# --------------------------------------------------------------------------- #
# tuple-level restatement (the device's input form, include/ecb.h)
# --------------------------------------------------------------------------- #
def tuples_valid(hapflag):
    """Filter on the packed flag word: BAM bits + host bit 12 (mate on another
    reference) + host bit 13 (next_pos < 0).  Same predicate as record_is_valid."""
    f = np.asarray(hapflag).astype(np.int64) & 0xFFFF
    paired = (f & 0x1) != 0
    bad_pair = ((f & 0x80) != 0) | ((f & 0x2) == 0) | ((f & 0x1000) != 0) | ((f & 0x2000) != 0)
    return ((f & 0x4) == 0) & ~(paired & bad_pair)


def ec_from_tuples(read_id, locus, hapflag, n_loci, n_haps, pos=None):
    """Slow literal EC build from device tuples: per read the set of (locus, hap),
    keyed, counted in first-appearance order; -> CSR A (bitmask values) and counts.
    Equivalent to scan()+build_incidence() when names map 1:1 to (locus, hap)."""
    read_id = np.asarray(read_id).astype(np.int64)
    locus = np.asarray(locus).astype(np.int64)
    hf = np.asarray(hapflag).astype(np.int64)
    valid = tuples_valid(hf)
    hap = (hf >> 16) & 0xFF
    ec = OrderedDict()
    cur, members = None, None
    first = []
    for i in np.nonzero(valid)[0]:
        rid = int(read_id[i])
        if rid != cur:
            if cur is not None:
                k = tuple(sorted(members))
                ec[k] = ec.get(k, 0) + 1
            cur, members = rid, set()
        members.add(int(locus[i]) * n_haps + int(hap[i]))
    if cur is not None:
        k = tuple(sorted(members))
        ec[k] = ec.get(k, 0) + 1
    indptr, indices, data = [0], [], []
    for k in ec:
        row = OrderedDict()
        for s in k:
            row[s // n_haps] = row.get(s // n_haps, 0) | (1 << (s % n_haps))
        indices += list(row.keys())
        data += list(row.values())
        indptr.append(len(indices))
    out = dict(indptr=np.array(indptr, dtype=np.int32), indices=np.array(indices, dtype=np.int32),
               data=np.array(data, dtype=np.int32), count=np.array(list(ec.values()), dtype=np.int32),
               n_all=len(read_id), n_valid=int(valid.sum()))
    if pos is not None:
        p = np.asarray(pos).astype(np.int64)
        slot = locus * n_haps + hap
        lo = np.full(n_loci * n_haps, np.iinfo(np.int64).max)
        hi = np.full(n_loci * n_haps, np.iinfo(np.int64).min)
        np.minimum.at(lo, slot[valid], p[valid])
        np.maximum.at(hi, slot[valid], p[valid])
        out["range"] = np.where(hi >= lo, hi - lo + 1, 0).reshape(n_loci, n_haps)
    return out
