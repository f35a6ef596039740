# -*- coding: utf-8 -*-
"""Drop-in for the reference's ``alntools/bam_utils_multisample.py``: a directory of BAM files, one EC x cell
count matrix (``convert`` keeps the signature of ``bam_utils_multisample.py:357``).

Split of the work:
* host -- what needs names: header maps, read runs and the cell of every read (field 14 of the ``'|||'``-separated
  read name, ``:270-280``), per file, with the reference's quirks kept: after the first read of a file the tracked
  name is the UNTRIMMED query name (``:292``), and the last read of every file is never counted (``:306-321``);
* device (libecb) -- everything read-sized: filter, target sets, ECs in first-appearance order, A, and the reduction
  of reads to distinct (EC, cell, file) triples with counts and first read index (``ec[key][cell] += 1``, ``:288-290``);
* device too (``ecb_ms_filter``): cell order (``cr_totals`` insertion order, ``:513-546``), the minimum-count filter and EC
  re-ranking (``:596-636``), N as CSC (``:737-791``), from the triples; the host only maps cell ids back to names.

File order: the reference takes ``glob.glob`` order (``:379``), which is filesystem-dependent and changes sample
order and EC order; ``convert`` here sorts the file names.  ``convert_files`` takes an explicit order.
"""
from __future__ import annotations

import glob
import os
import time

import numpy as np

from . import utils
from .bam_utils import BATCH_RECORDS, open_bam, write_range_file
from .bin_utils import ECMatrices, ecsave2
from .ecb import FLAG_MATE_OTHER_REF, FLAG_NEXT_POS_NEG, HAP_SHIFT, EcBuilder, EcbError
from .tuples import HeaderMaps, record_valid, trim_name

LOG = utils.get_logger()
CELL_BITS = 22


def scan_file(reader, maps):
    """One BAM file -> (tuples of its counted reads, cell name per counted read, counters, dropped-run records).

    Restates the name logic of ``bam_utils_multisample.py:257-300`` on the valid records of the file.
    """
    cols = {k: [] for k in ("flag", "tid", "pos", "ntid", "npos", "run")}
    run_cells = []
    tracked, run = None, -1
    n_all = n_valid = 0
    while hasattr(reader, "read_ms"):                                  # native decoder: the same rule, applied where the names are
        d, cells = reader.read_ms(BATCH_RECORDS)
        if d is None:
            c = {k: (np.concatenate(v) if v else np.zeros(0, dtype=np.int64)) for k, v in cols.items()}
            return c, c["run"] < run, run_cells[:max(run, 0)], dict(all=n_all, valid=n_valid)
        runs = run + np.cumsum(d["newrun"], dtype=np.int64)
        run = int(runs[-1])
        run_cells.extend(cells)
        n_all += len(runs)
        n_valid += int(d["valid"].sum())
        for k, v in (("flag", d["flag"]), ("tid", d["tid"]), ("pos", d["pos"]), ("ntid", d["next_tid"]), ("npos", d["next_pos"]), ("run", runs)):
            cols[k].append(np.array(v))
    while True:
        q, flag, tid, pos, ntid, npos = reader.read_batch(BATCH_RECORDS)
        if not q:
            break
        valid = record_valid(flag, tid.astype(np.int64), ntid.astype(np.int64), npos.astype(np.int64))
        runs = np.empty(len(q), dtype=np.int64)
        for i in range(len(q)):
            if valid[i]:
                name = q[i]
                if tracked is None:
                    tracked = trim_name(name)                         # :257-262
                    run, new = 0, True
                else:
                    new = False
                if tracked != trim_name(name):                        # :288
                    tracked = name                                    # untrimmed, :292
                    run += 1
                    new = True
                if new:
                    fields = tracked.split('|||')
                    if len(fields) < 15:
                        raise ValueError("read name %r has no cell id in '|||' field 14 (bam_utils_multisample.py:270-280)" % tracked)
                    run_cells.append(fields[14])
            runs[i] = run
        n_all += len(q)
        n_valid += int(valid.sum())
        for k, v in (("flag", flag), ("tid", tid), ("pos", pos), ("ntid", ntid), ("npos", npos), ("run", runs)):
            cols[k].append(v)
    c = {k: (np.concatenate(v) if v else np.zeros(0, dtype=np.int64)) for k, v in cols.items()}
    keep = c["run"] < run                                             # the file's last read is never counted (:306-321)
    return c, keep, run_cells[:max(run, 0)], dict(all=n_all, valid=n_valid)


def _tuples(c, sel, maps, read_base):
    tid = c["tid"][sel].astype(np.int64)
    flag = c["flag"][sel].astype(np.int64)
    ntid, npos = c["ntid"][sel].astype(np.int64), c["npos"][sel].astype(np.int64)
    valid = record_valid(flag, tid, ntid, npos)
    safe = np.where(valid, tid, 0)
    hostbits = np.where(tid != ntid, FLAG_MATE_OTHER_REF, 0) | np.where(npos < 0, FLAG_NEXT_POS_NEG, 0)
    hapflag = (flag & 0xFFF) | hostbits | (maps.tid2hap[safe].astype(np.int64) << HAP_SHIFT)
    rid = c["run"][sel] + read_base                                   # -1 (before the first read) wraps to 0xFFFFFFFF only at base 0
    rid = np.where(c["run"][sel] < 0, read_base - 1, rid)
    return ((rid & 0xFFFFFFFF).astype(np.uint32), maps.tid2locus[safe].astype(np.uint32), hapflag.astype(np.uint32),
            c["pos"][sel].astype(np.int32), valid)


def convert_files(bam_files, ec_filename, emase_filename, minimum_count=-1, range_filename=None, target_filename=None):
    """``bam_files`` in the given order -> ``.bin`` / ``.h5``; returns counters."""
    start_time = time.time()
    if not bam_files:
        raise ValueError("no bam files")
    LOG.info("Parsing the header of {}...".format(bam_files[0]))
    rd = open_bam(bam_files[0])
    targets = list(utils.parse_targets(target_filename).keys()) if target_filename else None
    maps = HeaderMaps(rd.references, rd.lengths, targets)              # header of the first file only (:399-465)
    rd.close()
    if len(bam_files) > (1 << (32 - CELL_BITS)):
        raise ValueError("more than %d input files" % (1 << (32 - CELL_BITS)))
    cell_ids = {}
    device = int(os.environ.get("ALNTOOLS_GPU", "0"))
    n_all = n_valid = 0
    track = range_filename is not None
    host_min = np.full((maps.n_loci, maps.n_haplotypes), np.iinfo(np.int32).max, dtype=np.int64)
    host_max = np.full((maps.n_loci, maps.n_haplotypes), np.iinfo(np.int32).min, dtype=np.int64)
    with EcBuilder(maps.n_loci, maps.n_haplotypes, device=device, track_ranges=track, multisample=True) as b:
        read_base = 0
        for fi, path in enumerate(bam_files):
            rd = open_bam(path, names=False, ms=True)
            c, keep, cells, ctr = scan_file(rd, maps)
            rd.close()
            n_all += ctr["all"]
            n_valid += ctr["valid"]
            rid, loc, hf, pos, _ = _tuples(c, keep, maps, read_base)
            if len(rid):
                b.push(rid, loc, hf, pos if track else None)
            if track and (~keep).any():                               # ranges see every valid alignment, counted or not (:249-253)
                _, l2, h2, p2, v2 = _tuples(c, ~keep, maps, 0)
                hap = (h2.astype(np.int64) >> HAP_SHIFT) & 0xFF
                np.minimum.at(host_min, (l2[v2].astype(np.int64), hap[v2]), p2[v2])
                np.maximum.at(host_max, (l2[v2].astype(np.int64), hap[v2]), p2[v2])
            if cells:
                ids = []
                for name in cells:
                    if name not in cell_ids:
                        cell_ids[name] = len(cell_ids)
                    ids.append(cell_ids[name])
                if len(cell_ids) > (1 << CELL_BITS):
                    raise ValueError("more than %d distinct cells" % (1 << CELL_BITS))
                b.push_cells(np.asarray(ids, dtype=np.uint32) | np.uint32(fi << CELL_BITS), read_base)
            read_base += len(cells)
        sizes = b.finalize()
        names = list(cell_ids.keys())
        LOG.info("Number of alignments: {:,}".format(n_valid))
        LOG.info("Number of main targets: {:,}".format(maps.n_loci))
        LOG.info("Number of haplotypes: {:,}".format(maps.n_haplotypes))
        LOG.info("Number of ECs: {:,}".format(sizes["n_ecs"]))
        LOG.info("Number of cells: {:,}".format(len(names)))
        try:
            f = b.ms_filter(len(names), minimum_count)                # cell order, filter, re-rank, N: on the device
        except EcbError as e:
            if e.code == -7:                                          # ECB_ERR_EMPTY
                raise ValueError("no cell reaches the minimum count")
            raise
        if track:
            mn, mx = b.export_range_minmax()
            mn, mx = np.minimum(mn.astype(np.int64), host_min), np.maximum(mx.astype(np.int64), host_max)
            write_range_file(range_filename, maps, np.where(mx >= mn, mx - mn + 1, 0))
    kept = [int(c) for c in f["kept_cells"]]
    n_ecs_kept = len(f["indptrA"]) - 1
    LOG.info("Number of ECs after filtering : {:,}".format(n_ecs_kept))
    LOG.info("Number of cells after filtering: {:,}".format(len(kept)))
    m = ECMatrices(maps.haplotypes, maps.main_targets, maps.lengths, [names[c] for c in kept],
                   f["indptrA"], f["indicesA"], f["dataA"], f["indptrN"], f["indicesN"], f["dataN"])
    if emase_filename:
        from . import emase_h5
        emase_h5.save(emase_filename, m, title='Multisample APM', incidence_only=False, count_2d=True, device=device)   # :806; N is a csc matrix whatever S is (:783-791)
    if ec_filename:
        try:
            os.remove(ec_filename)
        except OSError:
            pass
        ecsave2(ec_filename, m)
    LOG.info("Done, total time: {}".format(utils.format_time(start_time, time.time())))
    return dict(all_alignments=n_all, valid_alignments=n_valid, n_ecs=n_ecs_kept, n_cells=len(kept),
                n_ecs_before=sizes["n_ecs"], n_cells_before=len(names), samples=[names[c] for c in kept])


def convert(bam_filename, ec_filename, emase_filename, num_chunks=0, minimum_count=-1, number_processes=-1,
            temp_dir=None, range_filename=None, target_filename=None):
    """Same arguments as the reference (``bam_utils_multisample.py:357``); ``bam_filename`` is a directory."""
    if os.path.isfile(bam_filename):
        LOG.error('bam file must be a directory')
        return None
    bam_files = sorted(glob.glob(os.path.join(bam_filename, "*.bam")))
    if len(bam_files) == 0:
        LOG.error('No bam files found in directory: {}'.format(bam_filename))
        return None
    return convert_files(bam_files, ec_filename, emase_filename, minimum_count, range_filename, target_filename)
