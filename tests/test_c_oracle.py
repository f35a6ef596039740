"""The C restatement (oracle/ec_oracle.c) against the Python one and the reference's goldens."""
import json
import os

import numpy as np
import pytest

from alntools_amd import synth
from oracle import c_oracle
from oracle import ec_oracle as orc


@pytest.mark.parametrize("name,threads", [("g2_c1", 1), ("g2_c1", 3), ("g3_pe", 1), ("g3_pe", 8)])
def test_c_oracle_matches_python_oracle(golden_dir, name, threads):
    g = json.load(open(os.path.join(golden_dir, name + ".json")))
    spec = synth.SynthSpec(**g["spec"])
    t = synth.generate(spec, 0, spec.n_reads)
    a = c_oracle.ec_from_tuples(t["read_id"], t["locus"], t["hapflag"], spec.n_haps, threads=threads)
    b = orc.ec_from_tuples(t["read_id"], t["locus"], t["hapflag"], spec.n_loci, spec.n_haps)
    for k in ("indptr", "indices", "data", "count"):
        assert np.array_equal(a[k], b[k]), k
    assert a["n_valid"] == g["counters"]["# Valid Alignments"] and a["n_reads"] == g["n_reads"]
    assert len(a["count"]) == g["counters"]["# Equivalence Classes"]


def test_c_oracle_matches_reference_bin(golden_dir):
    spec = synth.SynthSpec(10000, 1000, 2)
    t = synth.generate(spec, 0, spec.n_reads)
    a = c_oracle.ec_from_tuples(t["read_id"], t["locus"], t["hapflag"], spec.n_haps, threads=4)
    w = orc.ecload_bytes(open(os.path.join(golden_dir, "g2_c1.bin"), "rb").read())
    for k, kk in (("indptr", "indptrA"), ("indices", "indicesA"), ("data", "dataA"), ("count", "dataN")):
        assert np.array_equal(a[k], w[kk]), k


def test_c_oracle_empty():
    with pytest.raises(ValueError):
        c_oracle.ec_from_tuples(np.array([0xFFFFFFFF], np.uint32), np.zeros(1, np.uint32), np.array([4], np.uint32), 2)
