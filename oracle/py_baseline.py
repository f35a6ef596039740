# -*- coding: utf-8 -*-
"""ORACLE -- test / measurement infrastructure, never the product path.

The reference's hot loops restated in plain Python ON THE TUPLE STREAM, for ``bench.py``'s ``cpu_baseline_py``: the same
per-alignment loop, string keys, ordered dicts, ordered merge and per-EC incidence build as ``alntools/bam_utils.py``
(scan ``:258-344``, merge ``:680-724``, A build ``:788-825``), run over contiguous read shards in ``P`` worker processes
as the reference does (``:646-680``, ``multiprocessing.Pool`` + ordered ``imap``).  Differences from the reference, all in
its favour: names are already integers here (``read_id``), so the per-record ``str.find`` / slice / compare of
``:289-306`` is an int compare, and nothing is decoded from BAM.  Parity: ``tests/test_oracle_golden.py`` holds it to
``ec_oracle.ec_from_tuples``.
"""
from __future__ import annotations

import multiprocessing as mp
import time
from collections import OrderedDict

import numpy as np

_G = {}


def scan_shard(bounds):
    """One worker: records [a, b) of the globals -> (OrderedDict key -> count, n_all, n_valid).  bam_utils.py:258-344."""
    a, b = bounds
    rid, loc, hf, n_haps = _G["rid"], _G["loc"], _G["hf"], _G["n_haps"]
    ec = OrderedDict()
    n_all = n_valid = 0
    cur = None
    reference_ids = []
    for i in range(a, b):
        n_all += 1
        f = int(hf[i])
        if f & 0x4:                                                      # :264-270
            continue
        if f & 0x1 and ((f & 0x80) or not (f & 0x2) or (f & 0x3000)):
            continue
        n_valid += 1
        reference_id = str(int(loc[i]) * n_haps + ((f >> 16) & 0xFF))    # the tid, as the string the reference keys on
        r = int(rid[i])
        if cur is None:
            cur = r
        if r != cur:                                                     # :306-320
            key = ",".join(sorted(reference_ids))
            ec[key] = ec.get(key, 0) + 1
            cur = r
            reference_ids = [reference_id]
        elif reference_id not in reference_ids:                          # :322-325
            reference_ids.append(reference_id)
    if reference_ids:
        key = ",".join(sorted(reference_ids))                            # :336-344
        ec[key] = ec.get(key, 0) + 1
    return ec, n_all, n_valid


def _init_worker(paths, n_haps):
    """Spawned worker: map the slice's arrays (written once by the parent) instead of inheriting them through a fork."""
    _G.update(rid=np.load(paths[0], mmap_mode="r"), loc=np.load(paths[1], mmap_mode="r"), hf=np.load(paths[2], mmap_mode="r"),
              n_haps=n_haps)


def _warm(_):
    return 0


def run(read_id, locus, hapflag, n_haps, processes, start_method="spawn"):
    """-> dict(n_ecs, nnz, n_all, n_valid, seconds_scan, seconds_merge, seconds_build).

    ``start_method="spawn"`` (default): the workers are fresh interpreters that map the arrays from files under /dev/shm (or the
    temp dir) -- the caller may hold a GPU context, which a forked child must not inherit.  Starting the pool is not timed (the
    reference's pool start is not what is being measured); ``seconds_scan`` is the ordered ``imap`` over the shards."""
    import os
    import tempfile
    rid = np.ascontiguousarray(read_id)
    n = len(rid)
    # contiguous shards cut at read boundaries (utils.partition + calculate_chunks cut at name changes, :1236-1247)
    cuts = [0]
    for k in range(1, processes):
        c = n * k // processes
        while 0 < c < n and rid[c] == rid[c - 1]:
            c += 1
        cuts.append(max(c, cuts[-1]))
    cuts.append(n)
    shards = [(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
    arrays = (rid, np.ascontiguousarray(locus), np.ascontiguousarray(hapflag))
    if processes > 1 and start_method != "fork":
        base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
        with tempfile.TemporaryDirectory(dir=base) as td:
            paths = [os.path.join(td, "%s.npy" % k) for k in ("rid", "loc", "hf")]
            for pth, a in zip(paths, arrays):
                np.save(pth, a)
            with mp.get_context(start_method).Pool(processes, initializer=_init_worker, initargs=(paths, n_haps)) as pool:
                pool.map(_warm, range(processes))                      # (every worker is up and has its arrays mapped)
                t0 = time.perf_counter()
                results = list(pool.imap(scan_shard, shards))
    else:
        _G.update(rid=arrays[0], loc=arrays[1], hf=arrays[2], n_haps=n_haps)
        t0 = time.perf_counter()
        if processes > 1:
            with mp.get_context("fork").Pool(processes) as pool:      # fork: the arrays are inherited, results are pickled back
                results = list(pool.imap(scan_shard, shards))
        else:
            results = [scan_shard(s) for s in shards]
    t1 = time.perf_counter()
    final = OrderedDict()                                              # :680-724
    n_all = n_valid = 0
    for ec, a, v in results:
        for k, c in ec.items():
            final[k] = final.get(k, 0) + c
        n_all += a
        n_valid += v
    t2 = time.perf_counter()
    indptr, indices, data = [0], [], []                                # :788-825, as rows of A (value = haplotype bitmask)
    for key in final:
        row = OrderedDict()
        for t in key.split(","):
            t = int(t)
            row[t // n_haps] = row.get(t // n_haps, 0) | (1 << (t % n_haps))
        for l in sorted(row):
            indices.append(l)
            data.append(row[l])
        indptr.append(len(indices))
    t3 = time.perf_counter()
    return dict(n_ecs=len(final), nnz=len(indices), n_all=n_all, n_valid=n_valid, count=list(final.values()),
                indptr=indptr, indices=indices, data=data,
                seconds_scan=t1 - t0, seconds_merge=t2 - t1, seconds_build=t3 - t2)
