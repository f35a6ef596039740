"""Host-side mirror (header maps, tuple encoding, .bin writer/reader) against the oracle and the reference's
goldens.  CPU only: nothing here touches libecb's compute entry points."""
import json
import os

import numpy as np
import pytest

from alntools_amd import bin_utils, synth, utils
from alntools_amd.tuples import HeaderMaps, TupleEncoder
from oracle import ec_oracle as orc


def test_header_maps_match_oracle(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "g1_edge.json")))
    refs = [r[0] for r in g["references"]]
    lens = [r[1] for r in g["references"]]
    for targets_txt in (None, g["targets_txt"]):
        o = orc.header_maps(refs, lens, targets_txt)
        t = list(orc.parse_targets(targets_txt).keys()) if targets_txt else None
        m = HeaderMaps(refs, lens, t)
        assert m.main_targets == list(o["main_targets"].keys())
        assert m.haplotypes == o["haplotypes"] == ['', 'A', 'B']
        assert np.array_equal(m.lengths, o["lengths"])
        for tid, name in enumerate(refs):
            target, hap = orc.split_reference_name(name)
            assert m.main_targets[m.tid2locus[tid]] == target and m.haplotypes[m.tid2hap[tid]] == hap


def test_non_bijective_header_is_rejected():
    with pytest.raises(ValueError):
        HeaderMaps(["A_", "A"], [10, 10])
    with pytest.raises(ValueError):
        HeaderMaps(["A_1", "A_1"], [10, 10])


@pytest.mark.parametrize("batch", [1, 3, 1000])
def test_tuple_encoder_reproduces_reference_bin_via_oracle(golden_dir, batch):
    """records -> tuples (this repo's host code) -> tuple-level oracle == the reference's .bin arrays"""
    g = json.load(open(os.path.join(golden_dir, "g1_edge.json")))
    refs = [r[0] for r in g["references"]]
    lens = [r[1] for r in g["references"]]
    recs = g["records"]
    m = HeaderMaps(refs, lens)
    enc = TupleEncoder(m)
    parts = []
    for a in range(0, len(recs), batch):
        chunk = recs[a:a + batch]
        cols = list(zip(*chunk))
        parts.append(enc.encode(list(cols[0]), np.array(cols[1]), np.array(cols[2]), np.array(cols[3]),
                                np.array(cols[4]), np.array(cols[5])))
    t = {k: np.concatenate([p[k] for p in parts]) for k in ("read_id", "locus", "hapflag", "pos")}
    # contract: non-decreasing, steps of at most one, steps only on valid records
    rid = t["read_id"].astype(np.int64)
    rid[rid == 0xFFFFFFFF] = -1
    assert np.all(np.diff(rid) >= 0) and np.all(np.diff(rid) <= 1)
    got = orc.ec_from_tuples(t["read_id"], t["locus"], t["hapflag"], m.n_loci, m.n_haplotypes, pos=t["pos"])
    w = orc.ecload_bytes(open(os.path.join(golden_dir, "g1_edge.bin"), "rb").read())
    for a, b in (("indptr", "indptrA"), ("indices", "indicesA"), ("data", "dataA"), ("count", "dataN")):
        assert np.array_equal(got[a], w[b]), a
    assert got["n_valid"] == g["counters"]["# Valid Alignments"]
    # and the range numbers are those of the reference's range file
    rows = open(os.path.join(golden_dir, "g1_edge.range.txt")).read().splitlines()[1:]
    ref_rng = np.array([[int(x) for x in r.split("\t")[1:]] for r in rows])
    assert np.array_equal(got["range"], ref_rng)


def test_bin_writer_is_byte_exact(golden_dir):
    for name in ("g1_edge.bin", "g1_edge_targets.bin", "g2_c1.bin", "g5_binwalk.bin", "g4_multi_min0.bin",
                 "g4_multi_min60.bin"):
        path = os.path.join(golden_dir, name)
        b = open(path, "rb").read()
        m = bin_utils.ecload(path)
        assert bin_utils.ecsave2_bytes(m) == b, name
    m = bin_utils.ecload(os.path.join(golden_dir, "g5_binwalk.bin"))
    g = json.load(open(os.path.join(golden_dir, "g5_binwalk.json")))["ecload"]
    for h, ref in enumerate(g["data"]):
        c = m.haplotype_csc(h)
        assert c.indptr.tolist() == ref["indptr"] and c.indices.tolist() == ref["indices"]


def test_synthetic_read_ids_follow_the_contract():
    for paired in (False, True):
        spec = synth.SynthSpec(3000, 200, 4, paired=paired)
        t = synth.generate(spec, 0, 3000)
        rid = t["read_id"].astype(np.int64)
        rid[rid == 0xFFFFFFFF] = -1
        d = np.diff(np.concatenate([[-1], rid]))
        assert np.all((d == 0) | (d == 1))
        assert np.all(orc.tuples_valid(t["hapflag"])[d == 1])
        assert rid[-1] + 1 == t["n_reads"]


def test_synthetic_loci_of_a_read_keep_the_stride_they_were_given():
    """``SynthSpec(locus_stride=...)``: the loci of a read are its base locus plus multiples of the stride (``bench.py --workload c3s / c3q``:
    the LDS table's bad cases); the default, 1 -- what every fixture holds -- gives runs of consecutive loci."""
    T = 5000
    for stride in (1, 4, 64):
        t = synth.generate(synth.SynthSpec(2000, T, 8, locus_stride=stride), 0, 2000)
        ok = orc.tuples_valid(t["hapflag"])
        seen = 0
        for r in np.unique(t["read_id"][ok])[:300]:
            loci = np.unique(t["locus"][ok & (t["read_id"] == r)]).astype(np.int64)
            if loci.max() - loci.min() > T // 2:                 # (a read that wraps around the last locus)
                continue
            assert set((loci - loci.min()) % stride) == {0} and loci.max() - loci.min() <= 4 * stride
            seen += len(loci) > 1
        assert seen > 50


def test_c_abi_library_loads_and_exports_every_declared_symbol():
    import re
    from alntools_amd import ecb
    lib = ecb.load()
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "ecb.h")).read()
    declared = set(re.findall(r"\b(ecb_[a-z_]+)\s*\(", hdr))
    assert declared == set(ecb.SYMBOLS), declared ^ set(ecb.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s
    assert lib.ecb_abi_version() == ecb.ABI_VERSION == 4
    # the host-side decoder library and its header
    from alntools_amd import bamdec
    bamdec.build()
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "bamdec.h")).read()
    declared = set(re.findall(r"\b(bd_[a-z_]+)\s*\(", hdr))
    assert declared == set(bamdec.SYMBOLS), declared ^ set(bamdec.SYMBOLS)
    for s in declared:
        assert hasattr(bamdec.lib(), s), s
    assert bamdec.lib().bd_abi_version() == bamdec.ABI_VERSION == 4


def test_plain_c_program_links_against_the_abi(tmp_path):
    import subprocess
    from alntools_amd import ecb
    ecb.load()
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.join(here, "..")
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.join(root, "alntools_amd")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-I", os.path.join(root, "include"), os.path.join(here, "abi_smoke.c"),
                           "-o", exe, "-L", libdir, "-l:libecb.so", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "abi ok" in out.stdout


def test_sanitized_host_build_passes_the_argument_checks(tmp_path):
    """libecb's host half under ASAN + UBSAN (python -m alntools_amd.build --asan): the plain-C ABI program -- bad struct size,
    too many haplotypes, create without a device (here) or a tiny run (on a GPU box) -- must come through without a report."""
    import subprocess
    from alntools_amd import build as b
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.join(here, "..")
    so = b.build_sanitized(out=str(tmp_path / "libecb_asan.so"))
    clang = os.path.join(os.path.dirname(os.path.realpath(b.HIPCC)), "..", "lib", "llvm", "bin", "clang")
    if not os.path.exists(clang):
        clang = "/opt/rocm/lib/llvm/bin/clang"
    rt = os.path.dirname(subprocess.check_output([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], text=True).strip())
    exe = str(tmp_path / "abi_asan")
    subprocess.check_call([clang, "-fsanitize=address,undefined", "-shared-libsan", "-std=c11", "-I", os.path.join(root, "include"),
                           os.path.join(here, "abi_smoke.c"), "-o", exe, "-L", str(tmp_path), "-l:libecb_asan.so",
                           "-Wl,-rpath," + str(tmp_path), "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath," + rt])
    import torch
    leaks = "0" if torch.cuda.is_available() else "1"      # (with a device the HIP runtime's own allocations outlive the program)
    out = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=%s:halt_on_error=1" % leaks,
                                                                      UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1"))
    assert out.returncode == 0 and "abi ok" in out.stdout and "ERROR" not in out.stderr and "runtime error" not in out.stderr, (out.returncode, out.stdout, out.stderr[-2000:])


def test_emase_h5_round_trip_through_libhdf5(golden_dir, tmp_path):
    """ec2emase / emase2ec (bin_utils.py:979-1028): .bin -> .h5 -> .bin is byte-identical; the .h5 holds the per-haplotype
    CSC matrices the reference stores (checked against scipy).  Needs only libhdf5 (ctypes)."""
    from alntools_amd import emase_h5
    try:
        emase_h5._backend()
    except RuntimeError:
        pytest.skip("no HDF5 library in this environment")
    for name in ("g2_c1.bin", "g1_edge.bin", "g4_multi_min0.bin"):
        src = os.path.join(golden_dir, name)
        h5, back = str(tmp_path / (name + ".h5")), str(tmp_path / (name + ".back"))
        bin_utils.ec2emase(src, h5, hapcsc=emase_h5.scipy_hapcsc)      # (no GPU here: the checker stands in for the device conversion)
        bin_utils.emase2ec(h5, back, csr=emase_h5.scipy_csr)
        assert open(back, "rb").read() == open(src, "rb").read(), name
    if emase_h5._backend() == "libhdf5":
        from alntools_amd import h5lite
        m = bin_utils.ecload(os.path.join(golden_dir, "g2_c1.bin"))
        with h5lite.File(str(tmp_path / "g2_c1.bin.h5")) as f:
            assert tuple(f.get_attr('/', 'shape')) == m.shape
            assert f.get_attr('/', 'mtype') == b'csc_matrix'
            assert list(f.get_attr('/', 'hname')) == m.hname
            for h in range(m.num_haplotypes):
                ref = m.haplotype_csc(h)
                assert np.array_equal(f.read_array('/h%d/indptr' % h), ref.indptr)
                assert np.array_equal(f.read_array('/h%d/indices' % h), ref.indices)
            assert np.array_equal(f.read_array('/count'), m.dataN.astype(np.float64))


def _h5dump_header(path):
    import shutil
    import subprocess
    exe = shutil.which("h5dump") or "/opt/conda/bin/h5dump"
    if not os.path.exists(exe):
        pytest.skip("no h5dump in this environment")
    out = subprocess.run([exe, "-H", "-A", path], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    return out.stdout


def test_emase_h5_structure_matches_the_documented_layout(golden_dir, tmp_path):
    """The groups, dataset types and attributes of the written .h5, read back by an independent tool (h5dump), against the
    layout the reference writes (Sparse3DMatrix.py:325-342, AlignmentPropertyMatrix.py:507-532): root attrs incidence_only /
    mtype / shape / hname, /h<i>/{indptr, indices} uint32 (+ data float64 unless incidence-only), /lengths, /lname, /rname,
    /sname, and /count as a float64 vector (ec2emase of a one-sample .bin) or as the uint32 group of a 2-D matrix
    (bam2emase: bam_utils.py:845 makes count a csc column; several samples)."""
    import re
    from alntools_amd import emase_h5
    try:
        emase_h5._backend()
    except RuntimeError:
        pytest.skip("no HDF5 library in this environment")
    m = bin_utils.ecload(os.path.join(golden_dir, "g2_c1.bin"))
    vec, grp = str(tmp_path / "vec.h5"), str(tmp_path / "grp.h5")
    emase_h5.save(vec, m, title="t", incidence_only=False, hapcsc=emase_h5.scipy_hapcsc)                 # what ec2emase writes
    emase_h5.save(grp, m, title="bam2ec", incidence_only=True, count_2d=True, hapcsc=emase_h5.scipy_hapcsc)   # what bam2emase writes

    def datasets(txt):
        out = {}
        for mm in re.finditer(r'DATASET "([^"]+)" \{\s*DATATYPE\s+(\S+)', txt):
            out[mm.group(1)] = mm.group(2)
        return out

    hv, hg = _h5dump_header(vec), _h5dump_header(grp)
    for txt in (hv, hg):
        for a in ("incidence_only", "mtype", "shape", "hname"):
            assert 'ATTRIBUTE "%s"' % a in txt, a
        for g in ["h%d" % h for h in range(m.num_haplotypes)]:
            assert 'GROUP "%s"' % g in txt
        assert txt.count('DATASET "indptr"') >= m.num_haplotypes and txt.count('DATASET "indices"') >= m.num_haplotypes
        for name in ("lengths", "lname", "rname", "sname"):
            assert 'DATASET "%s"' % name in txt, name
    dv, dg = datasets(hv), datasets(hg)
    assert dv["indptr"] == dv["indices"] == "H5T_STD_U32LE"                       # index_dtype='uint32' (Sparse3DMatrix.py:335-336)
    assert dv["data"] == "H5T_IEEE_F64LE" and dv["count"] == "H5T_IEEE_F64LE"     # data_dtype=float; count vector
    assert 'GROUP "count"' not in hv and 'GROUP "count"' in hg                    # vector vs 2-D count
    assert "data" in dg and dg["data"] == "H5T_STD_U32LE"                         # the only "data" left is /count/data, uint32 (APM.py:521)
    assert 'DATASET "count"' not in hg
    # and the two files read back to the same matrices
    for path in (vec, grp):
        back = emase_h5.load(path, csr=emase_h5.scipy_csr)
        assert np.array_equal(back.indptrA, m.indptrA) and np.array_equal(back.indicesA, m.indicesA) and np.array_equal(back.dataA, m.dataA)
        assert np.array_equal(back.dataN, m.dataN) and back.sname == m.sname and back.lname == m.lname


def test_emase_h5_pickled_attributes_unpickle_to_shape_and_haplotype_names(golden_dir, tmp_path):
    """The two attributes PyTables stores as pickled Python objects -- ``shape`` = (T, H, E) (Sparse3DMatrix.py:329) and ``hname`` =
    the haplotype list (AlignmentPropertyMatrix.py:511) -- read back as RAW bytes by an independent tool (h5dump) and un-pickled
    here: what the loader of the reference gets from ``get_node_attr`` (Sparse3DMatrix.py:36, AlignmentPropertyMatrix.py:124)."""
    import pickle
    import re
    import subprocess
    from alntools_amd import emase_h5
    try:
        emase_h5._backend()
    except RuntimeError:
        pytest.skip("no HDF5 library in this environment")
    exe = os.path.join(os.environ.get("CONDA_PREFIX", "/opt/conda"), "bin", "h5dump")
    if not os.path.exists(exe):
        pytest.skip("no h5dump in this environment")
    m = bin_utils.ecload(os.path.join(golden_dir, "g2_c1.bin"))
    path = str(tmp_path / "a.h5")
    emase_h5.save(path, m, title="t", incidence_only=True, count_2d=True, hapcsc=emase_h5.scipy_hapcsc)

    def raw_attr(name):
        out = subprocess.run([exe, "-e", "-a", "/" + name, path], capture_output=True, text=True)      # (-e: escapes, not raw newlines)
        assert out.returncode == 0, out.stderr
        data = out.stdout[out.stdout.index("DATA {"):]
        pieces = re.findall(r'"((?:[^"\\]|\\.)*)"', data)           # h5dump prints a long string in quoted pieces
        txt = "".join(pieces)
        return txt.encode("latin-1").decode("unicode_escape").encode("latin-1")

    assert tuple(int(x) for x in pickle.loads(raw_attr("shape"))) == m.shape
    hname = pickle.loads(raw_attr("hname"))
    assert [h.decode() if isinstance(h, bytes) else str(h) for h in hname] == list(m.hname)
    assert raw_attr("mtype") .rstrip(b"\0") == b"csc_matrix"        # (a plain string: the loader calls .decode('utf-8') on it, Sparse3DMatrix.py:54)


def test_open_bam_leaves_a_file_the_native_decoder_refuses_to_the_reference_reader(tmp_path, monkeypatch):
    """A SAM text file is not BGZF: ``bd_open`` refuses it (BD_ERR_FORMAT -> ValueError).  ``open_bam`` then falls through to
    pysam -- the reference's own reader (bam_utils.py:561), which opens SAM and CRAM too -- instead of failing on a file the reference takes."""
    import sys
    import types
    from alntools_amd import bam_utils, bamdec
    if not bamdec.available():
        pytest.skip("libbamdec.so not built")
    sam = tmp_path / "x.sam"
    sam.write_text("@HD\tVN:1.0\n@SQ\tSN:t1_A\tLN:100\nr1\t0\tt1_A\t1\t255\t4M\t*\t0\t0\tACGT\tIIII\n")
    with pytest.raises(ValueError):
        bamdec.NativeBamReader(str(sam))
    opened = []

    class FakeAlignmentFile(object):
        references, lengths = ("t1_A",), (100,)

        def __init__(self, name, check_sq=True):
            opened.append(name)

        def fetch(self, until_eof=False):
            return iter(())

        def close(self):
            pass

    monkeypatch.setitem(sys.modules, "pysam", types.SimpleNamespace(AlignmentFile=FakeAlignmentFile))
    rd = bam_utils.open_bam(str(sam), names=False)
    assert opened == [str(sam)] and type(rd).__name__ == "_PysamReader" and rd.references == ("t1_A",)


@pytest.mark.parametrize("batch", [1, 7, 100000])
def test_native_bam_decoder_yields_the_tuples_of_the_python_reader(golden_dir, tmp_path, batch):
    """csrc/bamdec.c (BGZF inflate on threads, record parse, filter verdict, read heads by name runs) against bamio.BamReader +
    TupleEncoder.encode on the edge-case fixture (names with spaces, re-appearing names, filtered records between a read's
    alignments) and on a multi-block paired-end file: same read ids, loci, haplotype/flag words and positions."""
    from alntools_amd import bamdec, bamio, bam_utils
    bamdec.build()
    g = json.load(open(os.path.join(golden_dir, "g1_edge.json")))
    cases = [("edge", [tuple(r) for r in g["references"]], [tuple(r) for r in g["records"]])]
    spec = synth.SynthSpec(4000, 300, 4, paired=True)
    cases.append(("pe", spec.references(), list(synth.raw_records(spec, 0, spec.n_reads))))
    for name, refs, recs in cases:
        bam = str(tmp_path / (name + ".bam"))
        bamio.write_bam(bam, refs, recs, level=1)
        py = bamio.BamReader(bam)
        nat = bamdec.NativeBamReader(bam, threads=3)
        assert nat.references == py.references and nat.lengths == py.lengths
        m = HeaderMaps(py.references, py.lengths)
        want, got = [], []
        old = bam_utils.BATCH_RECORDS
        bam_utils.BATCH_RECORDS = batch
        try:
            want = list(bam_utils.iter_tuple_batches(py, TupleEncoder(m)))
            got = [{k: np.array(v) for k, v in t.items()} for t in bam_utils.iter_tuple_batches(nat, TupleEncoder(m))]
        finally:
            bam_utils.BATCH_RECORDS = old
        py.close(); nat.close()
        # ... and through the decoded BAM fields + the encoder's own arithmetic (read_decoded / encode_decoded), the route of
        # encoders that are not the plain TupleEncoder
        nat = bamdec.NativeBamReader(bam, threads=2)
        enc2, fields = TupleEncoder(m), []
        while True:
            d = nat.read_decoded(batch)
            if d is None:
                break
            fields.append({k: np.array(v) for k, v in enc2.encode_decoded(**d).items()})
        nat.close()
        for k in ("read_id", "locus", "hapflag", "pos"):
            a = np.concatenate([t[k] for t in want]); b = np.concatenate([t[k] for t in got]); c = np.concatenate([t[k] for t in fields])
            assert len(a) == len(recs) and np.array_equal(a, b) and np.array_equal(a, c), (name, k)
            assert all(t[k].dtype == w[k].dtype for t, w in zip(got, want)), k
        assert sum(t["n_valid"] for t in want) == sum(t["n_valid"] for t in got) == sum(t["n_valid"] for t in fields)
    with pytest.raises((IOError, ValueError)):
        bamdec.NativeBamReader(os.path.join(golden_dir, "g1_edge.json"))


def test_native_multisample_scan_matches_the_python_scan(golden_dir, tmp_path):
    """csrc/bamdec.c:bd_read_ms (the multisample run rule -- tracked name cut at its first space only until the first switch,
    bam_utils_multisample.py:257-300 -- and the cell barcodes, field 14 of the tracked name) against the Python restatement in
    bam_utils_multisample.scan_file on the reference's multisample fixture (which has names with spaces) and on names built to
    hit the quirk in both states: same runs per record, same cells, same counters."""
    from alntools_amd import bamdec, bamio, bam_utils, bam_utils_multisample as bum
    bamdec.build()
    g = json.load(open(os.path.join(golden_dir, "g4_multi.json")))
    refs = [tuple(r) for r in g["references"]]
    files = {k: [tuple(r) for r in v] for k, v in g["files"].items()}
    cell = lambda c: "|||".join(["x"] * 14 + [c, "tail"])
    quirk = [(cell("C1") + " a", 0, 0, 10, -1, -1), (cell("C1") + " b", 0, 1, 11, -1, -1),      # first run: tracked is the cut name
             (cell("C2") + " a", 0, 0, 12, -1, -1), (cell("C2") + " a", 0, 1, 13, -1, -1),      # after a switch: tracked is the whole name
             (cell("C2"), 4, 0, 14, -1, -1), (cell("C3"), 0, 2, 15, -1, -1), (cell("C3"), 0, 1, 16, -1, -1),
             (" " + cell("C4"), 0, 0, 17, -1, -1), (" " + cell("C4"), 0, 1, 18, -1, -1), (cell("C5"), 0, 0, 19, -1, -1)]
    files["quirk.bam"] = quirk
    for name, recs in files.items():
        bam = str(tmp_path / name)
        bamio.write_bam(bam, refs, recs)
        for batch in (3, 100000):
            old = bam_utils.BATCH_RECORDS, bum.BATCH_RECORDS
            bam_utils.BATCH_RECORDS = bum.BATCH_RECORDS = batch
            try:
                py = bamio.BamReader(bam)
                want = bum.scan_file(py, None)
                py.close()
                nat = bamdec.NativeBamReader(bam, threads=2)
                got = bum.scan_file(nat, None)
                nat.close()
            finally:
                bam_utils.BATCH_RECORDS, bum.BATCH_RECORDS = old
            for k in ("flag", "tid", "pos", "ntid", "npos", "run"):
                assert np.array_equal(np.asarray(want[0][k], dtype=np.int64), np.asarray(got[0][k], dtype=np.int64)), (name, batch, k)
            assert np.array_equal(want[1], got[1]) and want[2] == got[2] and want[3] == got[3], (name, batch)
    bamio.write_bam(str(tmp_path / "nocell.bam"), refs, [("plain_name", 0, 0, 1, -1, -1)])
    with pytest.raises(ValueError):
        bum.scan_file(bamdec.NativeBamReader(str(tmp_path / "nocell.bam")), None)


def test_decoder_progress_follows_the_records_handed_out_not_the_read_ahead(tmp_path):
    """``bd_progress`` (what a multi-GPU deal-out cuts a file by): a file small enough to be read ahead whole must still report about
    half-way when half of its records have been handed out, and 1.0 at the end -- ABI 3 reported 1.0 from the first batch on, which dealt
    one read to every rank but the last."""
    from alntools_amd import bamdec, bamio
    bamdec.build()
    spec = synth.SynthSpec(6000, 300, 4, paired=True)
    recs = list(synth.raw_records(spec, 0, spec.n_reads))
    bam = str(tmp_path / "small.bam")
    bamio.write_bam(bam, spec.references(), recs, level=1)
    nat = bamdec.NativeBamReader(bam, threads=2)
    enc = TupleEncoder(HeaderMaps(nat.references, nat.lengths))
    seen, marks = 0, []
    while True:
        t = nat.read_tuples(len(recs) // 8 + 1, enc)
        if t is None:
            break
        seen += len(t["read_id"])
        marks.append((seen / float(len(recs)), nat.progress()))
    nat.close()
    assert len(marks) >= 8 and marks[-1][1] == 1.0
    assert all(b >= a for (_, a), (_, b) in zip(marks, marks[1:]))           # monotone
    assert all(abs(frac - prog) < 0.15 for frac, prog in marks), marks        # tracks the records handed out (by bytes: within a few blocks)


def test_rank_zero_deals_a_decoded_file_out_in_contiguous_read_ranges():
    """``bam_utils._deal_out`` (the multi-GPU convert: rank 0 decodes once): whatever the batch boundaries and wherever the decoder's
    progress crosses r / N, every rank is dealt whole reads, in order, numbered from 0, and the shards put back together are the stream."""
    from alntools_amd import bam_utils

    rng = np.random.RandomState(5)
    n_reads = 400
    lens = rng.randint(1, 9, n_reads)
    rid = np.repeat(np.arange(n_reads, dtype=np.uint32), lens)
    rid = np.concatenate([np.full(3, 0xFFFFFFFF, dtype=np.uint32), rid])          # records before the first read (invalid ones)
    n = len(rid)
    loc = rng.randint(0, 50, n).astype(np.uint32)
    hf = rng.randint(0, 1 << 20, n).astype(np.uint32)
    pos = rng.randint(0, 1000, n).astype(np.int32)

    for world, batch in ((3, 57), (4, 1), (2, n), (5, 7)):
        class Reader(object):
            at = 0

            def progress(self):
                return min(1.0, self.at / float(n))

        rd = Reader()

        def batches(reader, enc):
            for a in range(0, n, batch):
                reader.at = min(n, a + batch)
                yield dict(read_id=rid[a:a + batch], locus=loc[a:a + batch], hapflag=hf[a:a + batch], pos=pos[a:a + batch])

        got = {r: [] for r in range(world)}
        done = []

        class Builder(object):
            def push(self, r, l, h, p):
                got[0].append((np.array(r), np.array(l), np.array(h), np.array(p)))

        def send(dst, arrays):
            if arrays is None:
                done.append(dst)
            else:
                got[dst].append(tuple(np.array(a) for a in arrays))

        orig = bam_utils.iter_tuple_batches
        bam_utils.iter_tuple_batches = batches
        try:
            bam_utils._deal_out(rd, None, world, Builder(), True, send)
        finally:
            bam_utils.iter_tuple_batches = orig
        assert sorted(done) == list(range(1, world))
        back, base = [], 0
        for r in range(world):
            if not got[r]:
                continue
            cols = [np.concatenate([b[i] for b in got[r]]) for i in range(4)]
            local = cols[0].astype(np.int64)
            local[cols[0] == 0xFFFFFFFF] = -1
            if r:
                assert local[0] == 0 and local.min() == 0                      # a shard starts at a read boundary, its reads numbered from 0
            assert np.all(np.diff(local) >= 0) and np.all(np.diff(local) <= 1)
            glob = np.where(local < 0, np.int64(0xFFFFFFFF), local + base).astype(np.uint32)
            back.append((glob, cols[1], cols[2], cols[3]))
            base += int(local.max()) + 1
        for i, exp in enumerate((rid, loc, hf, pos)):
            assert np.array_equal(np.concatenate([b[i] for b in back]), exp), (world, batch, i)
        if batch < n // world:                                                   # (progress moves while the file is read: more than one rank gets reads)
            assert sum(1 for r in range(world) if got[r]) >= 2
