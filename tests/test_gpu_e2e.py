"""End to end on the GPU: BAM file -> alntools_amd.bam_utils.convert() -> .bin / range file, compared byte for
byte with what the reference wrote for the same BAM (tests/golden)."""
import hashlib
import json
import os

import pytest

from alntools_amd import bamio, methods, synth

pytestmark = pytest.mark.gpu


def _bytes(p):
    with open(p, "rb") as f:
        return f.read()


def test_edge_case_bam_matches_reference_bytes(golden_dir, tmp_path):
    g = json.load(open(os.path.join(golden_dir, "g1_edge.json")))
    bam = str(tmp_path / g["sample"])
    bamio.write_bam(bam, [tuple(r) for r in g["references"]], [tuple(r) for r in g["records"]])
    out, rng = str(tmp_path / "o.bin"), str(tmp_path / "o.range")
    sizes = methods.bam2ec(bam, out, range_filename=rng)
    assert _bytes(out) == _bytes(os.path.join(golden_dir, "g1_edge.bin"))
    assert open(rng).read() == open(os.path.join(golden_dir, "g1_edge.range.txt")).read()
    assert sizes["valid_alignments"] == g["counters"]["# Valid Alignments"]
    tf = str(tmp_path / "targets.txt")
    open(tf, "w").write(g["targets_txt"])
    methods.bam2ec(bam, out, range_filename=rng, target_filename=tf)
    assert _bytes(out) == _bytes(os.path.join(golden_dir, "g1_edge_targets.bin"))
    assert open(rng).read() == open(os.path.join(golden_dir, "g1_edge_targets.range.txt")).read()


def test_config1_bam_matches_reference_bytes(golden_dir, tmp_path):
    g = json.load(open(os.path.join(golden_dir, "g2_c1.json")))
    spec = synth.SynthSpec(**g["spec"])
    bam = str(tmp_path / g["sample"])
    bamio.write_bam(bam, spec.references(), synth.raw_records(spec, 0, spec.n_reads), level=1)
    out, rng = str(tmp_path / "c1.bin"), str(tmp_path / "c1.range")
    methods.bam2ec(bam, out, range_filename=rng)
    assert _bytes(out) == _bytes(os.path.join(golden_dir, "g2_c1.bin"))
    assert open(rng).read() == open(os.path.join(golden_dir, "g2_c1.range.txt")).read()


@pytest.mark.parametrize("name", ["g3_pe", "g3_mid"])
def test_8_haplotype_bams_match_reference_md5(golden_dir, tmp_path, name):
    g = json.load(open(os.path.join(golden_dir, name + ".json")))
    spec = synth.SynthSpec(**g["spec"])
    bam = str(tmp_path / g["sample"])
    bamio.write_bam(bam, spec.references(), synth.raw_records(spec, 0, spec.n_reads), level=1)
    out, rng = str(tmp_path / "o.bin"), str(tmp_path / "o.range")
    sizes = methods.bam2ec(bam, out, range_filename=rng)
    b = _bytes(out)
    assert len(b) == g["bin_len"] and hashlib.md5(b).hexdigest() == g["bin_md5"]
    assert hashlib.md5(open(rng).read().encode()).hexdigest() == g["range_md5"]
    assert sizes["n_ecs"] == g["counters"]["# Equivalence Classes"]


def test_no_valid_alignment_fails_cleanly(tmp_path):
    from alntools_amd.ecb import EcbError
    bam = str(tmp_path / "u.bam")
    bamio.write_bam(bam, [("T_A", 100)], [("r1", 4, -1, -1, -1, -1)])
    with pytest.raises(EcbError) as e:
        methods.bam2ec(bam, str(tmp_path / "u.bin"))
    assert e.value.code == -7


@pytest.mark.parametrize("mc,tag", [(-1, "0"), (20, "20"), (60, "60")])
def test_multisample_directory_matches_reference_bytes(golden_dir, tmp_path, mc, tag):
    from alntools_amd import bam_utils_multisample as ms
    g = json.load(open(os.path.join(golden_dir, "g4_multi.json")))
    refs = [tuple(r) for r in g["references"]]
    paths = []
    for fname in g["glob_order"]:                      # the order the reference's glob returned when the golden was made
        p = str(tmp_path / fname)
        bamio.write_bam(p, refs, [tuple(r) for r in g["files"][fname]])
        paths.append(p)
    out, rng = str(tmp_path / "m.bin"), str(tmp_path / "m.range")
    r = ms.convert_files(paths, out, None, minimum_count=mc, range_filename=rng)
    assert _bytes(out) == _bytes(os.path.join(golden_dir, "g4_multi_min%s.bin" % tag))
    c = g["counters"][str(mc)]
    assert r["valid_alignments"] == c["Number of alignments"]
    assert r["n_ecs"] == c["Number of ECs after filtering"] and r["n_cells"] == c["Number of cells after filtering"]
    assert r["n_ecs_before"] == c["Number of ECs"] and r["n_cells_before"] == c["Number of cells"]
    assert open(rng).read() == open(os.path.join(golden_dir, "g4_multi.range.txt")).read()


def test_command_line_bam2ec(golden_dir, tmp_path):
    from click.testing import CliRunner
    from alntools_amd.cli import cli
    g = json.load(open(os.path.join(golden_dir, "g1_edge.json")))
    bam = str(tmp_path / g["sample"])
    bamio.write_bam(bam, [tuple(r) for r in g["references"]], [tuple(r) for r in g["records"]])
    out = str(tmp_path / "cli.bin")
    r = CliRunner().invoke(cli, ["bam2ec", bam, out, "-v"])
    assert r.exit_code == 0, r.output
    assert _bytes(out) == _bytes(os.path.join(golden_dir, "g1_edge.bin"))


def test_plain_c_program_runs_the_hot_path(tmp_path):
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.join(here, "..")
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.join(root, "alntools_amd")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-I", os.path.join(root, "include"), os.path.join(here, "abi_smoke.c"),
                           "-o", exe, "-L", libdir, "-l:libecb.so", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "abi ok (device)" in out.stdout, (out.returncode, out.stdout, out.stderr)


def test_bam2emase_then_emase2ec_gives_the_reference_bin(golden_dir, tmp_path):
    """bam2emase writes the EMASE .h5 (through libhdf5); converting it back with emase2ec must give the very bytes the
    reference's bam2ec wrote for the same BAM.  (The .h5 container itself is not pinned: PyTables is not installed.)"""
    from alntools_amd import emase_h5
    try:
        emase_h5._backend()
    except RuntimeError:
        pytest.skip("no HDF5 library on this box")
    g = json.load(open(os.path.join(golden_dir, "g1_edge.json")))
    bam = str(tmp_path / g["sample"])
    bamio.write_bam(bam, [tuple(r) for r in g["references"]], [tuple(r) for r in g["records"]])
    h5, back = str(tmp_path / "o.h5"), str(tmp_path / "back.bin")
    methods.bam2emase(bam, h5)
    methods.emase2ec(h5, back)
    assert _bytes(back) == _bytes(os.path.join(golden_dir, "g1_edge.bin"))


def test_command_line_file_conversions_import_no_pytorch(golden_dir, tmp_path):
    """``alntools bam2ec`` / ``bam2emase`` / ``ec2emase`` / ``emase2ec`` on one GPU, each in a fresh interpreter with ``-X importtime``:
    the reference's bytes come out and ``torch`` is never imported -- everything goes through libecb's host-pointer entry points
    (``ecb_push``, ``ecb_csr_to_hapcsc``, ``ecb_hapcsc_to_csr``); the reference has no such dependency either (``setup.py:21-30``)."""
    import subprocess
    import sys
    from alntools_amd import emase_h5
    try:
        emase_h5._backend()
    except RuntimeError:
        pytest.skip("no HDF5 library on this box")
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    g = json.load(open(os.path.join(golden_dir, "g1_edge.json")))
    bam = str(tmp_path / g["sample"])
    bamio.write_bam(bam, [tuple(r) for r in g["references"]], [tuple(r) for r in g["records"]])
    b1, h5a, h5b, b2 = (str(tmp_path / n) for n in ("a.bin", "a.h5", "b.h5", "b.bin"))
    env = dict(os.environ)
    env.pop("ALNTOOLS_TORCH", None)
    env.pop("ALNTOOLS_GPUS", None)
    for args in (["bam2ec", bam, b1], ["bam2emase", bam, h5a], ["ec2emase", b1, h5b], ["emase2ec", h5b, b2]):
        r = subprocess.run([sys.executable, "-X", "importtime", "-m", "alntools_amd.cli"] + args, cwd=root, env=env, capture_output=True, text=True)
        assert r.returncode == 0, (args, r.stderr[-2000:])
        imported = [l.split("|")[-1].strip() for l in r.stderr.splitlines() if l.startswith("import time:")]
        assert "alntools_amd.ecb" in imported, args
        assert not any(m == "torch" or m.startswith("torch.") for m in imported), (args, [m for m in imported if m.startswith("torch")][:5])
    assert _bytes(b1) == _bytes(os.path.join(golden_dir, "g1_edge.bin"))
    assert _bytes(b2) == _bytes(b1)
