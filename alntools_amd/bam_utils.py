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
import time

import numpy as np

from . import bamio, utils
from .bin_utils import ECMatrices, ecsave2
from .ecb import EcBuilder
from .tuples import HeaderMaps, TupleEncoder

LOG = utils.get_logger()
BATCH_RECORDS = 1 << 20


def open_bam(filename):
    """pysam when installed (the reference's decoder), else the built-in reader; both yield
    ``(qname, flag, tid, pos, next_tid, next_pos)`` and expose ``.references`` / ``.lengths``."""
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


def write_range_file(range_filename, maps, range_len):
    """``bam_utils.py:735-766``: header ``#<TAB>haplotypes``, one row per main target, ``max-min+1`` or ``0``."""
    with open(range_filename, "w") as fw:
        fw.write("#\t" + "\t".join(maps.haplotypes) + "\n")
        for l, main_target in enumerate(maps.main_targets):
            fw.write(main_target + "\t" + "\t".join(str(int(v)) for v in range_len[l]) + "\n")


def stream_bam_to_builder(bam_filename, builder, maps=None, target_filename=None, track_ranges=False,
                          encoder_factory=TupleEncoder):
    """Decode one BAM on the host and push its tuples; returns ``(maps, encoder, n_records)``."""
    reader = open_bam(bam_filename)
    try:
        if maps is None:
            targets = list(utils.parse_targets(target_filename).keys()) if target_filename else None
            maps = HeaderMaps(reader.references, reader.lengths, targets)
        enc = encoder_factory(maps)
        if builder is None:
            return maps, enc, reader
        n_rec = 0
        while True:
            q, flag, tid, pos, ntid, npos = reader.read_batch(BATCH_RECORDS)
            if not q:
                break
            t = enc.encode(q, flag, tid, pos, ntid, npos)
            builder.push(t["read_id"], t["locus"], t["hapflag"], t["pos"] if track_ranges else None)
            n_rec += len(q)
        return maps, enc, n_rec
    finally:
        if builder is not None:
            reader.close()


def convert(bam_filename, ec_filename, emase_filename, num_chunks=0, number_processes=-1, temp_dir=None,
            range_filename=None, sample=None, target_filename=None):
    """BAM -> EC ``.bin`` and/or EMASE ``.h5`` (same arguments and outputs as ``bam_utils.convert``)."""
    start_time = time.time()
    if sample is None:
        sample = os.path.basename(bam_filename)                      # bam_utils.py:552-554
        LOG.info("Sample not supplied, using filename: {}".format(sample))
    elif isinstance(sample, bytes):
        sample = sample.decode('ascii', 'ignore')
    LOG.info("Parsing file information ...")
    reader = open_bam(bam_filename)
    targets = None
    if target_filename:
        targets = list(utils.parse_targets(target_filename).keys())
        if len(targets) == 0:
            raise ValueError("Unable to parse target file")
    maps = HeaderMaps(reader.references, reader.lengths, targets)
    device = int(os.environ.get("ALNTOOLS_GPU", "0"))
    temp_time = time.time()
    with EcBuilder(maps.n_loci, maps.n_haplotypes, device=device, track_ranges=range_filename is not None) as b:
        enc = TupleEncoder(maps)
        while True:
            q, flag, tid, pos, ntid, npos = reader.read_batch(BATCH_RECORDS)
            if not q:
                break
            t = enc.encode(q, flag, tid, pos, ntid, npos)
            b.push(t["read_id"], t["locus"], t["hapflag"], t["pos"] if range_filename else None)
        reader.close()
        sizes = b.finalize()
        out = b.export()
        LOG.info("All results combined in {}, total time: {}".format(utils.format_time(temp_time, time.time()),
                                                                     utils.format_time(start_time, time.time())))
        LOG.info("# Valid Alignments: {:,}".format(sizes["valid_alignments"]))
        LOG.info("# Main Targets: {:,}".format(maps.n_loci))
        LOG.info("# Haplotypes: {:,}".format(maps.n_haplotypes))
        LOG.info("# Equivalence Classes: {:,}".format(sizes["n_ecs"]))
        # the reference logs the number of distinct tracked names, which misses a trailing one-alignment
        # read (bam_utils.py:296-306); this is the number of reads actually counted
        LOG.info("# Unique Reads: {:,}".format(sizes["n_reads"]))
        if range_filename:
            write_range_file(range_filename, maps, b.export_ranges())
    m = ECMatrices(maps.haplotypes, maps.main_targets, maps.lengths, [sample], out["indptrA"], out["indicesA"],
                   out["dataA"], out["indptrN"], out["indicesN"], out["dataN"])
    if emase_filename:
        LOG.info("Saving to {}...".format(emase_filename))
        from . import emase_h5
        try:
            os.remove(emase_filename)
        except OSError:
            pass
        emase_h5.save(emase_filename, m, title='bam2ec', incidence_only=True, count_2d=True)     # bam_utils.py:845, 861: count is a csc column
    if ec_filename:
        LOG.info("Saving to {}...".format(ec_filename))
        try:
            os.remove(ec_filename)
        except OSError:
            pass
        ecsave2(ec_filename, m)
    LOG.info("Done, total time: {}".format(utils.format_time(start_time, time.time())))
    return sizes
