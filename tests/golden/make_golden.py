#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE.

Build-container only.  Imports the unmodified reference package from
``/root/reference`` (read-only) with the stand-in third-party modules of
``_standins/`` ahead of it on ``sys.path`` (pysam, Bio.bgzf, past, tables are
not installed here), feeds it BAM files written by ``alntools_amd.bamio`` and
records what it produced.  The fixtures hold data only: the decoded record
stream that went in and the bytes/counters that came out.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.json|.bin|.txt

Fixtures:
  g1_edge.*      hand-made edge cases (filter rules, name trimming, name re-appearance, duplicates,
                 haplotype '' / leading underscore, target file, range file)
  g2_c1.*        BASELINE config 1: 10k single-end reads, 2 haplotypes x 1k transcripts
  g3_mid.*       8 haplotypes x 2k transcripts, 60k reads (md5 + counters only)
  g3_pe.*        paired-end 8 x 500, 8k reads
  g4_multi.*     multisample directory (3 files, cells in '|||' field 14, min-count edge)
  g4b_multi.*    multisample at some size: 6 files, 21 000 reads, 260 cells (records regenerated from tests/golden/g4b_gen.py)
  g5_binwalk.*   tiny APM -> reference ecsave2 bytes and reference ecload arrays
  g6_utils.json  utils.partition / int_to_list / list_to_int truth tables
"""
from __future__ import print_function

import glob
import hashlib
import json
import logging
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(HERE, "_standins"))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402

from alntools_amd import bamio, synth  # noqa: E402

from alntools import bam_utils, bam_utils_multisample, bin_utils, utils  # noqa: E402  (the reference)
from alntools.matrix.AlignmentPropertyMatrix import AlignmentPropertyMatrix as RefAPM  # noqa: E402


class _Capture(logging.Handler):
    def __init__(self):
        logging.Handler.__init__(self)
        self.lines = []

    def emit(self, record):
        self.lines.append(record.getMessage())


def _counters(lines):
    out = {}
    for ln in lines:
        for key in ("# Valid Alignments", "# Main Targets", "# Haplotypes", "# Equivalence Classes",
                    "# Unique Reads", "Number of reads processed", "Number of alignments",
                    "Number of ECs after filtering ", "Number of cells after filtering",
                    "Number of ECs", "Number of cells"):
            if ln.startswith(key + ":"):
                out[key.strip()] = int(ln.split(":")[1].replace(",", "").strip())
                break
    return out


def run_reference(bam, target_file=None, want_range=True, multisample=False, minimum_count=-1):
    """-> (bin bytes, range text, counters, extras) from the reference's convert()."""
    tmp = tempfile.mkdtemp(prefix="golden_")
    cap = _Capture()
    root = logging.getLogger("alntools")
    root.addHandler(cap)
    root.setLevel(logging.INFO)
    try:
        ec = os.path.join(tmp, "out.bin")
        rng = os.path.join(tmp, "out.range") if want_range else None
        if multisample:
            bam_utils_multisample.convert(bam, ec, None, 0, minimum_count, 1, tmp, rng, target_file)
        else:
            bam_utils.convert(bam, ec, None, num_chunks=1, number_processes=1, temp_dir=tmp,
                              range_filename=rng, sample=None, target_filename=target_file)
        data = open(ec, "rb").read() if os.path.exists(ec) else None
        rtext = open(rng).read() if rng and os.path.exists(rng) else None
        return data, rtext, _counters(cap.lines), cap.lines
    finally:
        root.removeHandler(cap)
        shutil.rmtree(tmp, ignore_errors=True)


def md5(b):
    return hashlib.md5(b).hexdigest()


def dump_json(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=1, sort_keys=True)
        f.write("\n")


# --------------------------------------------------------------------------- #
def g1_edge():
    refs = [("T1_A", 1000), ("T2_A", 1500), ("T1_B", 1010), ("T2_B", 1490), ("T3", 700),
            ("_T4", 300), ("T5_B", 820), ("T6_A", 640), ("T6_B", 650)]
    U, P = 0x4, 0x1
    R1, R2, PP = 0x40, 0x80, 0x2
    recs = [
        # qname, flag, tid, pos, next_tid, next_pos
        ("lead_unmapped", U, -1, -1, -1, -1),
        ("rA", 0, 0, 10, -1, -1), ("rA", 0, 2, 12, -1, -1), ("rA", 16, 0, 30, -1, -1),   # dup tid 0
        ("rB", U, -1, -1, -1, -1),                                                        # invisible
        ("rA", 0, 1, 5, -1, -1),                                                          # still rA's run
        ("rC extra words", 0, 4, 100, -1, -1), ("rC other", 0, 5, 7, -1, -1),             # trimmed -> rC
        (" lead", 0, 6, 1, -1, -1), (" lead x", 0, 7, 2, -1, -1),                         # find(' ')==0: no trim
        ("rA", 0, 0, 900, -1, -1), ("rA", 0, 2, 0, -1, -1),                               # name re-appears
        ("pe1", P | PP | R1, 1, 50, 1, 200), ("pe1", P | PP | R2 | 0x10, 1, 200, 1, 50),
        ("pe1", P | PP | R1, 3, 55, 3, 210), ("pe1", P | PP | R2 | 0x10, 3, 210, 3, 55),
        ("pe2", P | R1, 0, 70, 0, 300),                                                   # improper
        ("pe2", P | PP | R1, 2, 71, 0, 300),                                              # mate elsewhere
        ("pe2", P | PP | R1, 7, 72, 7, -1),                                               # next_pos < 0
        ("pe2", P | PP | R1, 8, 73, 8, 400), ("pe2", P | PP | R2, 8, 400, 8, 73),
        ("pe2", P | PP | R1 | U, 8, 73, 8, 400),                                          # unmapped bit wins
        ("rD", 0, 8, 640, -1, -1), ("rD", 0, 7, 0, -1, -1), ("rD", 0, 4, 699, -1, -1),
        ("rE", 0, 2, 3, -1, -1), ("rE", 0, 0, 4, -1, -1),                                 # same EC as rA
        ("rF", 0, 7, 10, -1, -1), ("rF", 0, 8, 11, -1, -1), ("rF", 0, 4, 12, -1, -1),     # same EC as rD
        ("rG", 0, 5, 299, -1, -1),
        ("tail_unmapped", U, -1, -1, -1, -1),
        ("rH", 0, 6, 819, -1, -1),                                                        # trailing 1-alignment read
    ]
    targets_txt = "# comment line\nT9\tfoo\nT2 bar\n\tT1\n"
    work = tempfile.mkdtemp(prefix="g1_")
    try:
        bam = os.path.join(work, "edge.bam")
        bamio.write_bam(bam, refs, recs)
        tf = os.path.join(work, "targets.txt")
        open(tf, "w").write(targets_txt)
        b1, r1, c1, _ = run_reference(bam, None)
        b2, r2, c2, _ = run_reference(bam, tf)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    open(os.path.join(HERE, "g1_edge.bin"), "wb").write(b1)
    open(os.path.join(HERE, "g1_edge_targets.bin"), "wb").write(b2)
    open(os.path.join(HERE, "g1_edge.range.txt"), "w").write(r1)
    open(os.path.join(HERE, "g1_edge_targets.range.txt"), "w").write(r2)
    dump_json("g1_edge.json", dict(references=refs, records=recs, sample="edge.bam",
                                   targets_txt=targets_txt, counters=c1, counters_targets=c2,
                                   bin_md5=md5(b1), bin_targets_md5=md5(b2)))
    print("g1:", c1, len(b1), "bytes")


def _synth_case(name, spec, keep_bin):
    work = tempfile.mkdtemp(prefix=name + "_")
    try:
        bam = os.path.join(work, name + ".bam")
        bamio.write_bam(bam, spec.references(), synth.raw_records(spec, 0, spec.n_reads), level=1)
        b, r, c, _ = run_reference(bam, None)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    g = synth.generate(spec, 0, spec.n_reads, want_raw=True)
    stream_md5 = md5(b"".join(np.ascontiguousarray(g[k]).tobytes()
                              for k in ("read", "flag", "tid", "pos", "next_tid", "next_pos")))
    meta = dict(spec=dict(n_reads=spec.n_reads, n_loci=spec.n_loci, n_haps=spec.n_haps,
                          paired=spec.paired, seed=spec.seed, n_variants=spec.n_variants),
                gen_version=synth.GEN_VERSION, sample=name + ".bam", counters=c,
                n_records=g["n_records"], n_valid=g["n_valid"], n_reads=g["n_reads"],
                stream_md5=stream_md5, bin_md5=md5(b), bin_len=len(b), range_md5=md5(r.encode()))
    if keep_bin:
        open(os.path.join(HERE, name + ".bin"), "wb").write(b)
        open(os.path.join(HERE, name + ".range.txt"), "w").write(r)
    dump_json(name + ".json", meta)
    print(name + ":", c, len(b), "bytes")


def g2_c1():
    _synth_case("g2_c1", synth.SynthSpec(10000, 1000, 2), keep_bin=True)


def g3_mid():
    _synth_case("g3_mid", synth.SynthSpec(60000, 2000, 8), keep_bin=False)
    _synth_case("g3_pe", synth.SynthSpec(8000, 500, 8, paired=True), keep_bin=False)


def g4_multi():
    refs = [("G1_A", 900), ("G1_B", 905), ("G2_A", 400), ("G2_B", 410), ("G3_A", 1200), ("G3_B", 1190)]

    def qn(read, cell):
        f = ["x%d" % i for i in range(15)]
        f[0], f[14] = read, cell
        return "|||".join(f)

    rng = np.random.RandomState(7)
    files = {}
    for fi, fname in enumerate(("s_b.bam", "s_a.bam", "s_c.bam")):
        recs = []
        for r in range(120):
            cell = "CELL%02d" % int(rng.choice(6, p=[.35, .25, .2, .1, .07, .03]))
            name = qn("f%dr%03d" % (fi, r), cell)
            if rng.rand() < 0.05:
                recs.append((name, 4, -1, -1, -1, -1))
                continue
            g = int(rng.randint(3))
            tids = [2 * g] + ([2 * g + 1] if rng.rand() < 0.6 else [])
            if rng.rand() < 0.3:
                g2 = (g + 1) % 3
                tids.append(2 * g2 + int(rng.randint(2)))
            if rng.rand() < 0.1:
                tids.append(tids[0])
            for t in tids:
                recs.append((name, 0, t, int(rng.randint(refs[t][1])), -1, -1))
        if fi == 0:   # an EC seen only in a rare cell: dropped (and ECs re-ranked) once the cell is filtered
            for t in (0, 3, 4):
                recs.insert(30, (qn("f0rare", "RARE01"), 0, t, 5, -1, -1))
        if fi == 2:   # a name with a space: after a switch the reference tracks the UNTRIMMED name
            for t in (2, 3):
                recs.insert(40, (qn("f2space", "CELL03") + " tail", 0, t, 9, -1, -1))
        files[fname] = recs
    work = tempfile.mkdtemp(prefix="g4_")
    out = {}
    try:
        d = os.path.join(work, "bams")
        os.mkdir(d)
        for fname, recs in files.items():
            bamio.write_bam(os.path.join(d, fname), refs, recs)
        order = [os.path.basename(p) for p in glob.glob(os.path.join(d, "*.bam"))]
        for mc in (-1, 20, 60):
            b, r, c, lines = run_reference(d, None, multisample=True, minimum_count=mc)
            out[mc] = (b, r, c)
            open(os.path.join(HERE, "g4_multi_min%s.bin" % (mc if mc > 0 else "0")), "wb").write(b)
        open(os.path.join(HERE, "g4_multi.range.txt"), "w").write(out[-1][1])
    finally:
        shutil.rmtree(work, ignore_errors=True)
    dump_json("g4_multi.json", dict(references=refs, files=files, glob_order=order,
                                    counters={str(k): v[2] for k, v in out.items()},
                                    bin_md5={str(k): md5(v[0]) for k, v in out.items()}))
    print("g4:", {k: v[2] for k, v in out.items()}, "glob order", order)


def g4b_multi():
    """A multisample case of some size (6 files, 21 000 reads, 260 cells): pins the oracle -- and through it the numpy checker the
    full-size GPU tests use (tests/ms_checker.py) -- where g4's 360 reads and 8 cells cannot: hundreds of cells, most below any
    threshold, ECs that lose all their cells, cells that only appear in later files."""
    import g4b_gen
    refs = g4b_gen.references()
    files = g4b_gen.files()
    work = tempfile.mkdtemp(prefix="g4b_")
    out = {}
    try:
        d = os.path.join(work, "bams")
        os.mkdir(d)
        for fname, recs in files.items():
            bamio.write_bam(os.path.join(d, fname), refs, recs)
        order = [os.path.basename(p) for p in glob.glob(os.path.join(d, "*.bam"))]
        for mc in g4b_gen.MINCOUNTS:
            b, r, c, lines = run_reference(d, None, multisample=True, minimum_count=mc)
            out[mc] = (b, r, c)
            open(os.path.join(HERE, "g4b_multi_min%s.bin" % (mc if mc > 0 else "0")), "wb").write(b)
        open(os.path.join(HERE, "g4b_multi.range.txt"), "w").write(out[-1][1])
    finally:
        shutil.rmtree(work, ignore_errors=True)
    dump_json("g4b_multi.json", dict(generator="tests/golden/g4b_gen.py", seed=20260104, glob_order=order,
                                     counters={str(k): v[2] for k, v in out.items()},
                                     bin_md5={str(k): md5(v[0]) for k, v in out.items()}))
    print("g4b:", {k: v[2] for k, v in out.items()}, "glob order", order)


def g5_binwalk():
    from scipy.sparse import coo_matrix, csc_matrix
    apm = RefAPM(shape=(3, 2, 4), haplotype_names=["A", "B"], locus_names=["L0", "L1", "L2"],
                 read_names=np.arange(4).astype(str), sample_names=["S"])
    apm.lengths = np.array([[100, 101], [200, 0], [300, 303]], dtype=np.int32)
    #   EC0: L0 {A,B}; EC1: L1 {A}; EC2: L2 {B}; EC3: L0 {A}, L2 {A,B}
    apm.data[0] = coo_matrix((np.ones(4), ([0, 1, 3, 3], [0, 1, 0, 2])), shape=(4, 3))
    apm.data[1] = coo_matrix((np.ones(3), ([0, 2, 3], [0, 2, 2])), shape=(4, 3))
    apm.count = csc_matrix(np.matrix([5, 1, 7, 2]).T)
    apm.finalize()
    tmp = tempfile.mkdtemp(prefix="g5_")
    try:
        p = os.path.join(tmp, "w.bin")
        bin_utils.ecsave2(p, apm)
        b = open(p, "rb").read()
        back = bin_utils.ecload(p)
        loaded = dict(shape=list(back.shape), hname=list(back.hname), lname=list(back.lname),
                      sname=list(back.sname), lengths=np.asarray(back.lengths).tolist(),
                      count=np.asarray(back.count).tolist(),
                      data=[dict(indptr=m.indptr.tolist(), indices=m.indices.tolist(),
                                 data=m.data.tolist(), format=m.getformat()) for m in back.data])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    open(os.path.join(HERE, "g5_binwalk.bin"), "wb").write(b)
    dump_json("g5_binwalk.json", dict(bin_md5=md5(b), bin_len=len(b), ecload=loaded))
    print("g5:", len(b), "bytes")


def g6_utils():
    part = {}
    for n_items in (0, 1, 5, 7, 16, 1000):
        for n in (1, 2, 3, 4, 7, 8, 16):
            part["%d/%d" % (n_items, n)] = utils.partition(list(range(n_items)), n)
    i2l = {"%d/%d" % (c, s): utils.int_to_list(c, s) for c in (0, 1, 2, 3, 7, 121, 255) for s in (1, 3, 8)}
    l2i = {json.dumps(l): utils.list_to_int(l) for l in ([0, 0, 0], [0, 1, 0], [1, 1, 0], [1, 1, 1],
                                                        [1, 0, 0, 1, 1, 1, 1], [])}
    dump_json("g6_utils.json", dict(partition=part, int_to_list=i2l, list_to_int=l2i))
    print("g6 ok")


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g4b", "g5", "g6"]
    fns = dict(g1=g1_edge, g2=g2_c1, g3=g3_mid, g4=g4_multi, g4b=g4b_multi, g5=g5_binwalk, g6=g6_utils)
    for w in which:
        fns[w]()
