# -*- coding: utf-8 -*-
"""Drop-in for the reference's ``alntools/bam_utils.py`` hot path: ``convert()`` keeps its signature
and file side effects (``bam_utils.py:512``); the per-alignment worker, the merge and the A/N
construction (``:198-363, :680-724, :768-847``) run on the GPU through ``libecb.so``.

Not reproduced: the chunk-file plumbing that only exists to feed one pysam reader per process
(``chunk_bam_file``, ``calculate_chunks``, ``fix_bam``, temp BAMs) -- ``num_chunks``,
``number_processes`` and ``temp_dir`` are accepted and ignored.
"""
from __future__ import annotations

import os
import subprocess
import time

import numpy as np

from . import bamio, utils
from .bin_utils import ECMatrices, ecsave2
from .ecb import EcBuilder
from .tuples import HeaderMaps, TupleEncoder

LOG = utils.get_logger()
BATCH_RECORDS = 1 << 20


def open_bam(filename, names=True, ms=False):
    """``names=False`` (the single-sample path, which only needs to know where the read names CHANGE): the native decoder
    ``bamdec.NativeBamReader`` if ``libbamdec.so`` is built and ``ALNTOOLS_DECODER`` is not ``py`` (``ms``: for the multisample
    scan, which takes the cell barcodes out of the names in C as well).  Otherwise pysam when
    installed (the reference's decoder), else the built-in pure-Python reader; those yield
    ``(qname, flag, tid, pos, next_tid, next_pos)``.  All expose ``.references`` / ``.lengths``."""
    if not names and os.environ.get("ALNTOOLS_DECODER", "c") != "py":
        from . import bamdec
        if bamdec.available():
            try:
                return bamdec.NativeBamReader(filename)
            except ValueError:                       # not BGZF / not BAM (SAM, CRAM ...): what pysam, the reference's reader, may still open
                pass
            except (ImportError, OSError, subprocess.CalledProcessError) as e:      # a stale library that cannot be rebuilt here (no compiler, read-only tree)
                LOG.warning("libbamdec.so is not usable ({}): the file is decoded by the Python-level reader instead".format(e))
    try:
        import pysam
    except ImportError:
        return bamio.BamReader(filename)
    return _PysamReader(pysam.AlignmentFile(filename, check_sq=False))


class _PysamReader(object):
    def __init__(self, af):
        self._af = af
        self.references, self.lengths = tuple(af.references), tuple(af.lengths)
        self._it = af.fetch(until_eof=True)

    def read_batch(self, max_records):
        q, cols = [], []
        for a in self._it:
            q.append(a.query_name)
            cols.append((a.flag, a.reference_id, a.reference_start, a.next_reference_id, a.next_reference_start))
            if len(q) >= max_records:
                break
        c = np.asarray(cols, dtype=np.int64).reshape(-1, 5)
        return (q, c[:, 0].astype(np.uint16), c[:, 1].astype(np.int32), c[:, 2].astype(np.int32),
                c[:, 3].astype(np.int32), c[:, 4].astype(np.int32))

    def close(self):
        self._af.close()


def iter_tuple_batches(reader, enc):
    """Tuples of the whole file, a batch of records at a time, from either kind of reader (``open_bam``)."""
    if hasattr(reader, "read_tuples") and type(enc) is TupleEncoder and enc.trim == bool(reader.trim):
        while True:                                 # native decoder: the tuples come out of the C loop over the records
            t = reader.read_tuples(BATCH_RECORDS, enc)
            if t is None:
                return
            yield t
    if hasattr(reader, "read_decoded"):            # (an encoder of another kind: the decoded fields, then its own arithmetic)
        while True:
            d = reader.read_decoded(BATCH_RECORDS)
            if d is None:
                return
            yield enc.encode_decoded(**d)
    while True:
        q, flag, tid, pos, ntid, npos = reader.read_batch(BATCH_RECORDS)
        if not len(q):
            return
        yield enc.encode(q, flag, tid, pos, ntid, npos)


def write_range_file(range_filename, maps, range_len):
    """``bam_utils.py:735-766``: header ``#<TAB>haplotypes``, one row per main target, ``max-min+1`` or ``0``."""
    with open(range_filename, "w") as fw:
        fw.write("#\t" + "\t".join(maps.haplotypes) + "\n")
        for l, main_target in enumerate(maps.main_targets):
            fw.write(main_target + "\t" + "\t".join(str(int(v)) for v in range_len[l]) + "\n")


def stream_bam_to_builder(bam_filename, builder, maps=None, target_filename=None, track_ranges=False,
                          encoder_factory=TupleEncoder):
    """Decode one BAM on the host and push its tuples; returns ``(maps, encoder, n_records)``."""
    reader = open_bam(bam_filename, names=builder is None or encoder_factory is not TupleEncoder)
    try:
        if maps is None:
            targets = list(utils.parse_targets(target_filename).keys()) if target_filename else None
            maps = HeaderMaps(reader.references, reader.lengths, targets)
        enc = encoder_factory(maps)
        if builder is None:
            return maps, enc, reader
        n_rec = 0
        for t in iter_tuple_batches(reader, enc):
            builder.push(t["read_id"], t["locus"], t["hapflag"], t["pos"] if track_ranges else None)
            n_rec += len(t["read_id"])
        return maps, enc, n_rec
    finally:
        if builder is not None:
            reader.close()


def _write_outputs(maps, sample, sizes, out, range_len, ec_filename, emase_filename, range_filename, device=None):
    if range_filename:
        write_range_file(range_filename, maps, range_len)
    m = ECMatrices(maps.haplotypes, maps.main_targets, maps.lengths, [sample], out["indptrA"], out["indicesA"],
                   out["dataA"], out["indptrN"], out["indicesN"], out["dataN"])
    if emase_filename:
        LOG.info("Saving to {}...".format(emase_filename))
        from . import emase_h5
        try:
            os.remove(emase_filename)
        except OSError:
            pass
        emase_h5.save(emase_filename, m, title='bam2ec', incidence_only=True, count_2d=True, device=device)     # bam_utils.py:845, 861: count is a csc column
    if ec_filename:
        LOG.info("Saving to {}...".format(ec_filename))
        try:
            os.remove(ec_filename)
        except OSError:
            pass
        ecsave2(ec_filename, m)


def _log_summary(maps, sizes):
    LOG.info("# Valid Alignments: {:,}".format(sizes["valid_alignments"]))
    LOG.info("# Main Targets: {:,}".format(maps.n_loci))
    LOG.info("# Haplotypes: {:,}".format(maps.n_haplotypes))
    LOG.info("# Equivalence Classes: {:,}".format(sizes["n_ecs"]))
    # the reference logs the number of distinct tracked names, which misses a trailing one-alignment
    # read (bam_utils.py:296-306); this is the number of reads actually counted
    LOG.info("# Unique Reads: {:,}".format(sizes["n_reads"]))


def _deal_out(reader, enc, world, builder, track, group_send):
    """Rank 0 of a multi-GPU run: decode the file ONCE and deal its records out in contiguous read ranges -- the reference plans its
    chunks once and gives every process its own (``bam_utils.py:1174-1304, 646-680``).  Rank r is to have the reads between
    r / world and (r + 1) / world of the way through the file (by the decoder's ``progress()``: compressed bytes of the records
    parsed so far, no read-ahead counted); a batch that covers the stretch [p0, p1] of the file is cut where those marks fall
    within it, in proportion, at the next read that starts -- so a file of a single batch is shared out like a long one.
    Rank 0 pushes its own share as it goes.  ``group_send(dst, arrays or None)`` ships one batch (``None``: nothing more); what is
    shipped is always WHOLE reads -- the records of the read that is still open at the end of a batch wait for the next one -- so
    that a receiver may push a message from where it lands (``ecb_push_device`` takes whole reads).
    A reader that cannot say how far it is (the pysam fall-back) leaves everything with rank 0: said in the log.
    -> records dealt to each rank."""
    owner, base, dealt = 0, 0, 0                      # the rank being fed, the run counter of its first read, records it has had
    prev = 0xFFFFFFFF                                 # run counter of the record before the batch
    prog = getattr(reader, "progress", None)
    per_rank = [0] * world
    p0 = 0.0
    held = None                                       # the open read of the rank being fed (its columns), not shipped yet

    def ship(dst, part, last):
        """``part``: the next records of rank ``dst`` (columns, read ids local); ``last``: nothing of this rank's follows."""
        nonlocal held
        if held is not None:
            part = [np.concatenate([h, c]) for h, c in zip(held, part)]
            held = None
        if not last and len(part[0]):
            rid_ = part[0]
            k = len(rid_) - 1                         # first record of the trailing run of equal read ids (invalid records carry the id before them)
            tail = rid_[k]
            while k > 0 and rid_[k - 1] == tail:
                k -= 1
            held = [c[k:] for c in part]
            part = [c[:k] for c in part]
        if len(part[0]):
            group_send(dst, part)

    for t in iter_tuple_batches(reader, enc):
        rid = t["read_id"]
        n = len(rid)
        p1 = max(p0, min(1.0, float(prog()))) if prog else 0.0
        cols = [rid, t["locus"], t["hapflag"]] + ([t["pos"]] if track else [])
        before = np.concatenate([np.asarray([prev], dtype=rid.dtype), rid[:-1]])
        starts = np.flatnonzero((rid != before) & (rid != 0xFFFFFFFF))      # records that start a read
        lo = 0
        while True:
            cut = n
            mark = (owner + 1) / float(world)
            if owner < world - 1 and mark <= p1:     # the owner's stretch ends within this batch (or did before it, with no read starting since)
                at = lo if mark <= p0 else max(lo, int(n * (mark - p0) / max(p1 - p0, 1e-12)))
                cand = starts[starts >= at] if (dealt or at > lo) else starts[starts > lo]      # (a rank that has had nothing keeps what starts here)
                if len(cand):
                    cut = int(cand[0])
            if cut > lo:
                part = [c[lo:cut] for c in cols]
                local = np.where(part[0] == 0xFFFFFFFF, np.uint32(0xFFFFFFFF), part[0] - np.uint32(base)).astype(np.uint32)
                if owner == 0:
                    builder.push(local, part[1], part[2], part[3] if track else None)
                else:
                    ship(owner, [local] + part[1:], cut < n)      # (cut < n: the next read belongs to the next rank)
                dealt += cut - lo
                per_rank[owner] += cut - lo
            if cut == n:
                break
            if held is not None:                     # (nothing of this rank's in the batch, but its open read ends here)
                group_send(owner, held)
                held = None
            owner += 1                               # (the read at `cut` is the new owner's read 0)
            base = int(rid[cut])
            lo, dealt = cut, 0
        prev = int(rid[-1])
        p0 = p1
    if held is not None:                              # the stream ended: the open read is complete
        group_send(owner, held)
        held = None
    for r in range(1, world):
        group_send(r, None)
    if world > 1 and min(per_rank) * 4 * world < sum(per_rank):
        LOG.warning("uneven deal-out over {} ranks (records per rank: {}){}".format(
            world, per_rank, "" if prog else ": this reader does not report its progress, every record stayed with rank 0"))
    return per_rank


def _rank_convert(rank, world, port, backend, devices, bam_filename, ec_filename, emase_filename, range_filename, sample,
                  target_filename, result_path):
    """One process per GPU (the reference: one process per contiguous chunk range, ``bam_utils.py:646-680``).  Rank 0 decodes the
    BAM once and deals the records out in contiguous read ranges (:func:`_deal_out`; the tuples travel over the process group --
    xGMI under RCCL); every rank builds the EC table of its range on its GPU; the tables are merged by key range
    (``dist.exchange_and_merge``: the ordered merge of ``:680-724``), every rank finalizes its range, rank 0 puts the ranges
    together and writes."""
    import json
    import torch
    import torch.distributed as tdist
    from . import dist as ecdist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev_index = devices[rank]
    device = torch.device("cuda", dev_index)
    torch.cuda.set_device(device)
    # (a rank waits in its first receive until the decoder has reached its share of the file: minutes on a large BAM -- the
    #  process group's default timeout, ten minutes under nccl, would abort the job)
    import datetime
    patience = datetime.timedelta(seconds=int(os.environ.get("ALNTOOLS_DIST_TIMEOUT_S", str(24 * 3600))))
    if backend == "nccl":
        tdist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=patience)
    else:
        tdist.init_process_group(backend, rank=rank, world_size=world, timeout=patience)
    wire = device if backend == "nccl" else torch.device("cpu")        # where a message lives while it travels
    targets = list(utils.parse_targets(target_filename).keys()) if target_filename else None
    track = range_filename is not None
    n_cols = 4 if track else 3
    # every rank reads the header (the maps); only rank 0 reads on
    reader = open_bam(bam_filename, names=False)
    maps = HeaderMaps(reader.references, reader.lengths, targets)
    b = EcBuilder(maps.n_loci, maps.n_haplotypes, device=dev_index, track_ranges=track)
    if rank == 0:
        def send(dst, arrays):
            n = 0 if arrays is None else len(arrays[0])
            tdist.send(torch.tensor([n, 0 if arrays is None else 1], dtype=torch.int64, device=wire), dst)
            if n:       # one message: the columns back to back, each padded to whole 16 bytes (what ecb_push_device asks of a stream)
                ns = (n + 3) & ~3
                host = np.zeros(ns * len(arrays), dtype=np.int32)
                for i, a in enumerate(arrays):
                    host[i * ns:i * ns + n] = np.ascontiguousarray(a).view(np.int32)
                tdist.send(torch.from_numpy(host).to(wire), dst)
        _deal_out(reader, TupleEncoder(maps), world, b, track, send)
        reader.close()
    else:
        reader.close()
        while True:
            head = torch.empty(2, dtype=torch.int64, device=wire)
            tdist.recv(head, 0)
            n, more = (int(x) for x in head.cpu().tolist())
            if not more:
                break
            ns = (n + 3) & ~3
            msg = torch.empty(ns * n_cols, dtype=torch.int32, device=wire)
            tdist.recv(msg, 0)
            if wire.type == "cuda":                   # over RCCL the tuples arrive in HBM: pushed from where they are
                torch.cuda.current_stream(device).synchronize()      # (libecb works on its own stream)
                cols = [msg[i * ns:i * ns + n] for i in range(n_cols)]
                b.push_device(cols[0], cols[1], cols[2], cols[3] if track else None)
            else:
                a = msg.numpy()
                cols = [a[i * ns:i * ns + n] for i in range(n_cols)]
                b.push(cols[0].view(np.uint32), cols[1].view(np.uint32), cols[2].view(np.uint32), cols[3] if track else None)
    wrap = lambda e: e if backend == "nccl" else ecdist.HostStagedEngine(e)
    eng = wrap(ecdist.GpuEngine(b, device))
    fresh = lambda: wrap(ecdist.GpuEngine(EcBuilder(maps.n_loci, maps.n_haplotypes, device=dev_index, track_ranges=False), device))
    merged = ecdist.exchange_and_merge(eng, fresh, fresh, root=0, finalize_ranges=True)     # (every rank ranks and emits its own key range)
    range_len = None
    if track:
        mn, mx = b.export_range_minmax()
        range_len = ecdist.reduce_ranges(mn, mx, device=device if backend == "nccl" else None)
    if rank == 0:
        mb = merged.b
        sizes = mb.finalize()
        out = mb.export()
        _log_summary(maps, sizes)
        _write_outputs(maps, sample, sizes, out, range_len, ec_filename, emase_filename, range_filename, device=dev_index)
        with open(result_path, "w") as f:
            json.dump(sizes, f)
    tdist.barrier()
    tdist.destroy_process_group()


def _convert_multi(n_gpus, bam_filename, ec_filename, emase_filename, range_filename, sample, target_filename):
    """``ALNTOOLS_GPUS=N``: N processes, one per GPU, RCCL merge (``ALNTOOLS_DIST_BACKEND=gloo`` + ``ALNTOOLS_GPU_LIST=0,0``
    rehearses the same protocol with several ranks on one GPU, tables staged through host memory)."""
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
        result = os.path.join(td, "sizes.json")
        mp.spawn(_rank_convert, args=(n_gpus, port, backend, devices, bam_filename, ec_filename, emase_filename, range_filename,
                                      sample, target_filename, result), nprocs=n_gpus, join=True)
        return json.load(open(result))


def convert(bam_filename, ec_filename, emase_filename, num_chunks=0, number_processes=-1, temp_dir=None,
            range_filename=None, sample=None, target_filename=None):
    """BAM -> EC ``.bin`` and/or EMASE ``.h5`` (same arguments and outputs as ``bam_utils.convert``)."""
    start_time = time.time()
    if sample is None:
        sample = os.path.basename(bam_filename)                      # bam_utils.py:552-554
        LOG.info("Sample not supplied, using filename: {}".format(sample))
    elif isinstance(sample, bytes):
        sample = sample.decode('ascii', 'ignore')
    n_gpus = int(os.environ.get("ALNTOOLS_GPUS", "1"))
    if n_gpus > 1:
        sizes = _convert_multi(n_gpus, bam_filename, ec_filename, emase_filename, range_filename, sample, target_filename)
        LOG.info("Done, total time: {}".format(utils.format_time(start_time, time.time())))
        return sizes
    LOG.info("Parsing file information ...")
    reader = open_bam(bam_filename, names=False)
    targets = None
    if target_filename:
        targets = list(utils.parse_targets(target_filename).keys())
        if len(targets) == 0:
            raise ValueError("Unable to parse target file")
    maps = HeaderMaps(reader.references, reader.lengths, targets)
    device = int(os.environ.get("ALNTOOLS_GPU", "0"))
    temp_time = time.time()
    with EcBuilder(maps.n_loci, maps.n_haplotypes, device=device, track_ranges=range_filename is not None,
                   verify=bool(int(os.environ.get("ALNTOOLS_VERIFY", "0")))) as b:
        enc = TupleEncoder(maps)
        for t in iter_tuple_batches(reader, enc):
            b.push(t["read_id"], t["locus"], t["hapflag"], t["pos"] if range_filename else None)
        reader.close()
        sizes = b.finalize()
        out = b.export()
        LOG.info("All results combined in {}, total time: {}".format(utils.format_time(temp_time, time.time()),
                                                                     utils.format_time(start_time, time.time())))
        _log_summary(maps, sizes)
        range_len = b.export_ranges() if range_filename else None
    _write_outputs(maps, sample, sizes, out, range_len, ec_filename, emase_filename, range_filename, device=device)
    LOG.info("Done, total time: {}".format(utils.format_time(start_time, time.time())))
    return sizes
