"""The oracle (oracle/ec_oracle.py) against fixtures produced by running the reference
(tests/golden/make_golden.py).  CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest

from alntools_amd import synth
from oracle import ec_oracle as orc


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def _bytes(golden_dir, name):
    with open(os.path.join(golden_dir, name), "rb") as f:
        return f.read()


def _md5(b):
    return hashlib.md5(b).hexdigest()


def test_g1_edge_cases(golden_dir):
    g = _load(golden_dir, "g1_edge.json")
    refs = [r[0] for r in g["references"]]
    lens = [r[1] for r in g["references"]]
    recs = [tuple(r) for r in g["records"]]
    out = orc.convert_records(refs, lens, recs, g["sample"], None, want_range=True)
    assert out["bin"] == _bytes(golden_dir, "g1_edge.bin")
    assert out["range"] == open(os.path.join(golden_dir, "g1_edge.range.txt")).read()
    assert out["counters"]["valid"] == g["counters"]["# Valid Alignments"]
    assert out["counters"]["ecs"] == g["counters"]["# Equivalence Classes"]
    out = orc.convert_records(refs, lens, recs, g["sample"], g["targets_txt"], want_range=True)
    assert out["bin"] == _bytes(golden_dir, "g1_edge_targets.bin")
    assert out["range"] == open(os.path.join(golden_dir, "g1_edge_targets.range.txt")).read()


def test_unique_reads_log_quirk(golden_dir):
    # the reference's "# Unique Reads" misses a trailing one-alignment read
    g = _load(golden_dir, "g1_edge.json")
    r = orc.scan([tuple(x) for x in g["records"]])
    assert r["unique_reads"] == g["counters"]["# Unique Reads"]


def _synth_case(golden_dir, name):
    g = _load(golden_dir, name + ".json")
    assert g["gen_version"] == synth.GEN_VERSION
    spec = synth.SynthSpec(**g["spec"])
    s = synth.generate(spec, 0, spec.n_reads, want_raw=True)
    stream = b"".join(np.ascontiguousarray(s[k]).tobytes()
                      for k in ("read", "flag", "tid", "pos", "next_tid", "next_pos"))
    assert _md5(stream) == g["stream_md5"], "synthetic generator drifted from the fixture"
    refs = spec.references()
    out = orc.convert_records([r[0] for r in refs], [r[1] for r in refs],
                              synth.raw_records(spec, 0, spec.n_reads), g["sample"], None, want_range=True)
    assert out["counters"]["valid"] == g["counters"]["# Valid Alignments"]
    assert out["counters"]["ecs"] == g["counters"]["# Equivalence Classes"]
    assert len(out["bin"]) == g["bin_len"]
    assert _md5(out["bin"]) == g["bin_md5"]
    assert _md5(out["range"].encode()) == g["range_md5"]
    return spec, s, out


def test_g2_config1(golden_dir):
    spec, s, out = _synth_case(golden_dir, "g2_c1")
    assert out["bin"] == _bytes(golden_dir, "g2_c1.bin")
    # the tuple-level restatement agrees with the literal one
    t = orc.ec_from_tuples(s["read_id"], s["locus"], s["hapflag"], spec.n_loci, spec.n_haps, pos=s["pos"])
    w = orc.ecload_bytes(out["bin"])
    for a, b in (("indptr", "indptrA"), ("indices", "indicesA"), ("data", "dataA"), ("count", "dataN")):
        assert np.array_equal(t[a], w[b]), a
    assert t["n_valid"] == out["counters"]["valid"]


def test_g3_pe(golden_dir):
    spec, s, out = _synth_case(golden_dir, "g3_pe")
    t = orc.ec_from_tuples(s["read_id"], s["locus"], s["hapflag"], spec.n_loci, spec.n_haps)
    w = orc.ecload_bytes(out["bin"])
    for a, b in (("indptr", "indptrA"), ("indices", "indicesA"), ("data", "dataA"), ("count", "dataN")):
        assert np.array_equal(t[a], w[b]), a


@pytest.mark.slow
def test_g3_mid(golden_dir):
    _synth_case(golden_dir, "g3_mid")


def test_g4_multisample(golden_dir):
    g = _load(golden_dir, "g4_multi.json")
    refs = [r[0] for r in g["references"]]
    lens = [r[1] for r in g["references"]]
    files = [[tuple(r) for r in g["files"][f]] for f in g["glob_order"]]
    for mc, tag in ((-1, "0"), (20, "20"), (60, "60")):
        out = orc.convert_multisample(refs, lens, files, mc, None, want_range=True)
        assert out["bin"] == _bytes(golden_dir, "g4_multi_min%s.bin" % tag), mc
        c = g["counters"][str(mc)]
        assert out["counters"]["valid"] == c["Number of alignments"]
        assert out["counters"]["ecs"] == c["Number of ECs after filtering"]
        assert out["counters"]["cells"] == c["Number of cells after filtering"]
    assert out["range"] == open(os.path.join(golden_dir, "g4_multi.range.txt")).read()


def test_g5_bin_walk(golden_dir):
    g = _load(golden_dir, "g5_binwalk.json")
    b = _bytes(golden_dir, "g5_binwalk.bin")
    w = orc.ecload_bytes(b)
    assert w["indptrA"].tolist() == [0, 1, 2, 3, 5]
    assert w["indicesA"].tolist() == [0, 1, 2, 0, 2]
    assert w["dataA"].tolist() == [3, 1, 2, 1, 3]
    assert w["indptrN"].tolist() == [0, 4] and w["dataN"].tolist() == [5, 1, 7, 2]
    ref = g["ecload"]
    assert w["hname"] == ref["hname"] and w["lname"] == ref["lname"] and w["sname"] == ref["sname"]
    assert w["lengths"].tolist() == [[int(x) for x in row] for row in ref["lengths"]]
    # haplotype h of the reference's ecload == bit h of A
    from scipy.sparse import csr_matrix
    for h, m in enumerate(ref["data"]):
        bit = (w["dataA"] >> h) & 1
        mine = csr_matrix((bit, w["indicesA"].copy(), w["indptrA"].copy()), shape=(4, 3))
        mine.eliminate_zeros()
        mine = mine.tocsc()
        assert mine.indptr.tolist() == m["indptr"] and mine.indices.tolist() == m["indices"]


def test_g6_utils(golden_dir):
    g = _load(golden_dir, "g6_utils.json")
    for k, v in g["partition"].items():
        n_items, n = map(int, k.split("/"))
        assert orc.partition(list(range(n_items)), n) == v
    for k, v in g["int_to_list"].items():
        c, s = map(int, k.split("/"))
        assert orc.int_to_list(c, s) == v
    for k, v in g["list_to_int"].items():
        assert orc.list_to_int(json.loads(k)) == v


def test_no_valid_alignments_is_an_error():
    with pytest.raises(ValueError):
        orc.scan([("u", 4, -1, -1, -1, -1)])
