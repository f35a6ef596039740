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


def _push_files(b, maps, files, track, cell_of):
    """Scan ``files`` = [(global file index, path), ...] in order and push their counted reads into ``b`` (read ids local to this
    handle, from 0).  ``cell_of(name) -> id`` hands out cell ids.  -> (n_all, n_valid, reads pushed, host-side range extremes of
    the reads the reference never counts -- the last read of every file, ``:306-321`` -- which its ranges still see, ``:249-253``)."""
    n_all = n_valid = 0
    host_min = np.full((maps.n_loci, maps.n_haplotypes), np.iinfo(np.int32).max, dtype=np.int64)
    host_max = np.full((maps.n_loci, maps.n_haplotypes), np.iinfo(np.int32).min, dtype=np.int64)
    read_base = 0
    for fi, path in files:
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
            ids = np.asarray([cell_of(name) for name in cells], dtype=np.uint32)
            b.push_cells(ids | np.uint32(fi << CELL_BITS), read_base)
        read_base += len(cells)
    return n_all, n_valid, read_base, host_min, host_max


def _finish(maps, names, sizes, f, range_len, ec_filename, emase_filename, range_filename, device, n_all, n_valid, start_time):
    """The host's part after ``ecb_ms_filter``: names for the kept cells, the files."""
    if range_filename:
        write_range_file(range_filename, maps, range_len)
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


def _log_counts(maps, n_valid, sizes, n_cells):
    LOG.info("Number of alignments: {:,}".format(n_valid))
    LOG.info("Number of main targets: {:,}".format(maps.n_loci))
    LOG.info("Number of haplotypes: {:,}".format(maps.n_haplotypes))
    LOG.info("Number of ECs: {:,}".format(sizes["n_ecs"]))
    LOG.info("Number of cells: {:,}".format(n_cells))


def _filter(b, n_cells, minimum_count):
    try:
        return b.ms_filter(n_cells, minimum_count)                    # cell order, filter, re-rank, N: on the device
    except EcbError as e:
        if e.code == -7:                                              # ECB_ERR_EMPTY
            raise ValueError("no cell reaches the minimum count")
        raise


def _header_maps(bam_files, target_filename):
    LOG.info("Parsing the header of {}...".format(bam_files[0]))
    rd = open_bam(bam_files[0])
    targets = list(utils.parse_targets(target_filename).keys()) if target_filename else None
    maps = HeaderMaps(rd.references, rd.lengths, targets)              # header of the first file only (:399-465)
    rd.close()
    return maps


def convert_files(bam_files, ec_filename, emase_filename, minimum_count=-1, range_filename=None, target_filename=None):
    """``bam_files`` in the given order -> ``.bin`` / ``.h5``; returns counters.  ``ALNTOOLS_GPUS=N``: the files are dealt out to
    N processes, one per GPU (the reference: one worker per file, ``bam_utils_multisample.py:473-480``)."""
    start_time = time.time()
    if not bam_files:
        raise ValueError("no bam files")
    if len(bam_files) > (1 << (32 - CELL_BITS)):
        raise ValueError("more than %d input files" % (1 << (32 - CELL_BITS)))
    n_gpus = int(os.environ.get("ALNTOOLS_GPUS", "1"))
    if n_gpus > 1:
        return _convert_multi(n_gpus, bam_files, ec_filename, emase_filename, minimum_count, range_filename, target_filename)
    maps = _header_maps(bam_files, target_filename)
    cell_ids = {}

    def cell_of(name):
        if name not in cell_ids:
            if len(cell_ids) >= (1 << CELL_BITS):
                raise ValueError("more than %d distinct cells" % (1 << CELL_BITS))
            cell_ids[name] = len(cell_ids)
        return cell_ids[name]

    device = int(os.environ.get("ALNTOOLS_GPU", "0"))
    track = range_filename is not None
    with EcBuilder(maps.n_loci, maps.n_haplotypes, device=device, track_ranges=track, multisample=True) as b:
        n_all, n_valid, _, host_min, host_max = _push_files(b, maps, list(enumerate(bam_files)), track, cell_of)
        sizes = b.finalize()
        names = list(cell_ids.keys())
        _log_counts(maps, n_valid, sizes, len(names))
        f = _filter(b, len(names), minimum_count)
        range_len = None
        if track:
            mn, mx = b.export_range_minmax()
            mn, mx = np.minimum(mn.astype(np.int64), host_min), np.maximum(mx.astype(np.int64), host_max)
            range_len = np.where(mx >= mn, mx - mn + 1, 0)
    return _finish(maps, names, sizes, f, range_len, ec_filename, emase_filename, range_filename, device, n_all, n_valid, start_time)


def _rank_convert(rank, world, port, backend, devices, bam_files, ec_filename, emase_filename, minimum_count, range_filename,
                  target_filename, result_path):
    """One process per GPU.  Rank r takes the files ``[r F / world, (r + 1) F / world)`` -- contiguous, so the run's read order is
    the reference's file order -- and builds the EC table of their reads on its GPU; cell ids are agreed on first (names in the
    order the files bring them up: rank 0's cells first, then what rank 1 adds ...: the single process's ids); the tables are
    merged by key range over the process group (``dist.exchange_and_merge``), every rank finalizes the range it merged and rank 0
    places the rows, every rank reduces its reads to (EC, cell, file) triples against the merged ECs
    (``dist.exchange_multisample``), and rank 0 filters and writes."""
    import json
    import pickle
    import torch
    import torch.distributed as tdist
    from . import dist as ecdist
    start_time = time.time()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev_index = devices[rank]
    device = torch.device("cuda", dev_index)
    torch.cuda.set_device(device)
    import datetime
    patience = datetime.timedelta(seconds=int(os.environ.get("ALNTOOLS_DIST_TIMEOUT_S", str(24 * 3600))))      # (ranks scan files of very different sizes)
    if backend == "nccl":
        tdist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=patience)
    else:
        tdist.init_process_group(backend, rank=rank, world_size=world, timeout=patience)
    red = device if backend == "nccl" else torch.device("cpu")
    maps = _header_maps(bam_files, target_filename)
    F = len(bam_files)
    mine = [(fi, bam_files[fi]) for fi in range(rank * F // world, (rank + 1) * F // world)]
    track = range_filename is not None
    local_names = {}

    def cell_of(name):                                                # (local ids for now: this rank's order of first appearance)
        if name not in local_names:
            local_names[name] = len(local_names)
        return local_names[name]

    # Cell ids must be the run's before the reads' cells go to the device: scan first for names only?  No -- the cells of a read
    # are pushed with local ids and translated on the device side of the wire: push_cells takes ids, so the files are scanned
    # into host arrays of local ids per file, the name lists are exchanged, and the ids are mapped before push_cells.
    b = EcBuilder(maps.n_loci, maps.n_haplotypes, device=dev_index, track_ranges=track, multisample=True)
    pending = []                                                      # (first read, local cell ids | file << CELL_BITS) per file

    class _Deferred(object):                                          # the builder as _push_files sees it: cells wait for their ids
        def push(self, *a):
            b.push(*a)

        def push_cells(self, meta, first_read):
            pending.append((first_read, meta))

    n_all, n_valid, n_reads, host_min, host_max = _push_files(_Deferred(), maps, mine, track, cell_of)
    # the run's cell ids: names in rank order, each rank's in its own order of first appearance
    blob = pickle.dumps(list(local_names.keys()))
    sz = torch.tensor([len(blob)], dtype=torch.int64, device=red)
    allsz = torch.empty(world, dtype=torch.int64, device=red)
    tdist.all_gather_into_tensor(allsz, sz)
    allsz = [int(x) for x in allsz.cpu().tolist()]
    buf = torch.zeros(max(allsz) or 1, dtype=torch.uint8, device=red)
    buf[:len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(red)
    allbuf = torch.empty(world * buf.numel(), dtype=torch.uint8, device=red)
    tdist.all_gather_into_tensor(allbuf, buf)
    allbuf = allbuf.cpu().numpy().reshape(world, -1)
    cell_ids = {}
    for r in range(world):
        for name in pickle.loads(allbuf[r, :allsz[r]].tobytes()):
            if name not in cell_ids:
                cell_ids[name] = len(cell_ids)
    if len(cell_ids) > (1 << CELL_BITS):
        raise ValueError("more than %d distinct cells" % (1 << CELL_BITS))
    names = list(cell_ids.keys())
    remap = np.asarray([cell_ids[n] for n in local_names.keys()], dtype=np.uint32) if local_names else np.zeros(0, np.uint32)
    for first_read, meta in pending:
        b.push_cells(remap[meta & np.uint32((1 << CELL_BITS) - 1)] | (meta & ~np.uint32((1 << CELL_BITS) - 1)), first_read)
    wrap = lambda e: e if backend == "nccl" else ecdist.HostStagedEngine(e)
    eng = wrap(ecdist.GpuEngine(b, device))
    plain = lambda: wrap(ecdist.GpuEngine(EcBuilder(maps.n_loci, maps.n_haplotypes, device=dev_index), device))
    root = lambda: wrap(ecdist.GpuEngine(EcBuilder(maps.n_loci, maps.n_haplotypes, device=dev_index, multisample=True), device))
    # Every rank ranks and emits the key range it merged; the root only places the rows (``ecb_assemble_ranges_device``) and takes the
    # ECs' hashes off the finished rows for the second exchange -- it no longer adopts every merged table and finalizes alone.
    # (ALNTOOLS_DIST_FINALIZE=root: the earlier ending, the merged tables adopted by the root.)
    per_range = os.environ.get("ALNTOOLS_DIST_FINALIZE", "ranges") != "root"
    merged = ecdist.exchange_and_merge(eng, plain, root, root=0, finalize_ranges=per_range)
    sizes, n_ecs = None, 0
    if rank == 0:
        sizes = merged.b.finalize()
        n_ecs = sizes["n_ecs"]
    ecdist.exchange_multisample(eng, merged, n_ecs, root=0)
    tot = torch.tensor([n_all, n_valid], dtype=torch.int64, device=red)
    tdist.all_reduce(tot)
    n_all, n_valid = (int(x) for x in tot.cpu().tolist())
    range_len = None
    if track:
        mn, mx = b.export_range_minmax()
        mn, mx = np.minimum(mn.astype(np.int64), host_min), np.maximum(mx.astype(np.int64), host_max)
        range_len = ecdist.reduce_ranges(mn.astype(np.int32), mx.astype(np.int32), device=device if backend == "nccl" else None)
    if rank == 0:
        _log_counts(maps, n_valid, sizes, len(names))
        f = _filter(merged.b, len(names), minimum_count)
        out = _finish(maps, names, sizes, f, range_len, ec_filename, emase_filename, range_filename, dev_index, n_all, n_valid, start_time)
        with open(result_path, "w") as fh:
            json.dump(out, fh)
    tdist.barrier()
    tdist.destroy_process_group()


def _convert_multi(n_gpus, bam_files, ec_filename, emase_filename, minimum_count, range_filename, target_filename):
    """``ALNTOOLS_GPUS=N``: N processes, one per GPU (``ALNTOOLS_DIST_BACKEND=gloo`` + ``ALNTOOLS_GPU_LIST=0,0`` rehearses the same
    protocol with several ranks on one GPU, tables staged through host memory)."""
    import json
    import socket
    import tempfile
    import torch.multiprocessing as mp
    backend = os.environ.get("ALNTOOLS_DIST_BACKEND", "nccl")
    devices = [int(x) for x in os.environ.get("ALNTOOLS_GPU_LIST", ",".join(str(i) for i in range(n_gpus))).split(",")]
    if len(devices) != n_gpus:
        raise ValueError("ALNTOOLS_GPU_LIST names %d devices for ALNTOOLS_GPUS=%d" % (len(devices), n_gpus))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    with tempfile.TemporaryDirectory() as td:
        result = os.path.join(td, "out.json")
        mp.spawn(_rank_convert, args=(n_gpus, port, backend, devices, list(bam_files), ec_filename, emase_filename, minimum_count,
                                      range_filename, target_filename, result), nprocs=n_gpus, join=True)
        return json.load(open(result))


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
