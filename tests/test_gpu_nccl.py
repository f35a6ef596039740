"""The multi-GPU protocol over RCCL itself (backend ``nccl``), one process per GPU: runs by itself wherever two GPUs are
visible and is skipped on a one-GPU box (where tests/test_gpu_dist.py rehearses the same protocol over gloo with the ranks
sharing the card).  The reference's counterpart is the ordered merge of its workers' results (``bam_utils.py:646-724``)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank: fewer than two GPUs are visible")]

SPEC = dict(n_reads=120000, n_loci=5000, n_haps=8, paired=True)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _meta(g):
    g = g.astype(np.uint64)
    return (((g * np.uint64(2654435761)) % np.uint64(101)) | ((g % np.uint64(3)) << np.uint64(22))).astype(np.uint32)


def _worker(rank, world, port, out_path, per_range, multisample):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from alntools_amd import dist as ecdist
    from alntools_amd import ecb, synth
    spec = synth.SynthSpec(**SPEC)
    R = spec.n_reads
    a, z = rank * R // world, (rank + 1) * R // world
    base = synth.generate(spec, 0, a)["n_reads"] if a else 0
    t = synth.generate(spec, a, z, device=dev)
    b = ecb.EcBuilder(spec.n_loci, spec.n_haps, device=rank, ec_capacity=1 << 12, multisample=multisample)
    b.push_device(t["read_id"], t["locus"], t["hapflag"])
    if multisample:
        b.push_cells(_meta(np.arange(base, base + t["n_reads"])), 0)
    eng = ecdist.GpuEngine(b, dev)
    fresh = lambda ms: (lambda: ecdist.GpuEngine(ecb.EcBuilder(spec.n_loci, spec.n_haps, device=rank, ec_capacity=1 << 12, multisample=ms), dev))
    merged = ecdist.exchange_and_merge(eng, fresh(False), fresh(multisample), root=0, finalize_ranges=per_range)
    n_ecs, s = None, None
    if rank == 0:
        s = merged.b.finalize()
        n_ecs = s["n_ecs"]
    extra = {}
    if multisample:
        nt = ecdist.exchange_multisample(eng, merged, n_ecs, root=0)
        if rank == 0:
            extra = {"p_" + k: v for k, v in merged.b.export_pairs().items()}
            extra["nt"] = nt
    if rank == 0:
        np.savez(out_path, n_ecs=n_ecs, n_reads=s["n_reads"], all=s["all_alignments"], valid=s["valid_alignments"], **extra, **merged.b.export())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("per_range,multisample", [(True, False), (False, False), (False, True)])
def test_exchange_and_merge_over_rccl_equals_one_handle(tmp_path, per_range, multisample):
    """Two ranks, two GPUs, device tensors over RCCL point-to-point: key ranges exchanged and merged, finalized per range
    (single-sample) or adopted by the root (+ the multisample triples) == one handle over the whole stream, bit for bit."""
    from alntools_amd import ecb, synth
    out = str(tmp_path / "merged.npz")
    mp.spawn(_worker, args=(2, _free_port(), out, per_range, multisample), nprocs=2, join=True)
    got = np.load(out)
    spec = synth.SynthSpec(**SPEC)
    whole = synth.generate(spec, 0, spec.n_reads, device=torch.device("cuda:0"))
    with ecb.EcBuilder(spec.n_loci, spec.n_haps, multisample=multisample) as one:
        one.push_device(whole["read_id"], whole["locus"], whole["hapflag"])
        if multisample:
            one.push_cells(_meta(np.arange(whole["n_reads"])), 0)
        s = one.finalize()
        exp = one.export()
        exp_p = one.export_pairs() if multisample else None
    assert int(got["n_ecs"]) == s["n_ecs"] and int(got["n_reads"]) == s["n_reads"]
    assert int(got["all"]) == s["all_alignments"] and int(got["valid"]) == s["valid_alignments"]
    for k in ("indptrA", "indicesA", "dataA") + (() if multisample else ("dataN",)):
        assert np.array_equal(got[k], exp[k]), k
    if multisample:
        assert int(got["nt"]) == s["nnz_n"]
        for k in ("ec", "cell", "file", "count", "first"):
            assert np.array_equal(got["p_" + k], exp_p[k]), k


def test_convert_over_two_gpus_writes_the_reference_bytes(tmp_path, monkeypatch):
    """``ALNTOOLS_GPUS=2`` with the default backend (nccl = RCCL): convert() spawns one process per GPU, rank 0 decodes and deals
    the records out as device tensors, the receiver pushes them where they land.  Same bytes as the reference's goldens."""
    import json
    from alntools_amd import bam_utils, bamio, synth
    golden = os.path.join(os.path.dirname(__file__), "golden")
    monkeypatch.setenv("ALNTOOLS_GPUS", "2")
    monkeypatch.delenv("ALNTOOLS_DIST_BACKEND", raising=False)
    monkeypatch.delenv("ALNTOOLS_GPU_LIST", raising=False)
    for name in ("g1_edge", "g2_c1"):
        g = json.load(open(os.path.join(golden, name + ".json")))
        bam = str(tmp_path / g["sample"])
        if "records" in g:
            bamio.write_bam(bam, [tuple(r) for r in g["references"]], [tuple(r) for r in g["records"]])
        else:
            spec = synth.SynthSpec(**g["spec"])
            bamio.write_bam(bam, spec.references(), synth.raw_records(spec, 0, spec.n_reads), level=1)
        out, rng = str(tmp_path / (name + ".bin")), str(tmp_path / (name + ".range"))
        sizes = bam_utils.convert(bam, out, None, range_filename=rng)
        assert open(out, "rb").read() == open(os.path.join(golden, name + ".bin"), "rb").read(), name
        exp_rng = os.path.join(golden, name + ".range.txt")
        if os.path.exists(exp_rng):
            assert open(rng).read() == open(exp_rng).read(), name
        assert sizes["n_ecs"] > 0


def test_multisample_convert_over_two_gpus_writes_the_reference_bytes(tmp_path, monkeypatch):
    """``bam_utils_multisample.convert`` over two GPUs and RCCL: the reference's ``.bin`` bytes at its three thresholds."""
    import json
    from alntools_amd import bam_utils_multisample as ms, bamio
    golden = os.path.join(os.path.dirname(__file__), "golden")
    monkeypatch.setenv("ALNTOOLS_GPUS", "2")
    monkeypatch.delenv("ALNTOOLS_DIST_BACKEND", raising=False)
    monkeypatch.delenv("ALNTOOLS_GPU_LIST", raising=False)
    g = json.load(open(os.path.join(golden, "g4_multi.json")))
    refs = [tuple(r) for r in g["references"]]
    paths = []
    for fname in g["glob_order"]:
        p = str(tmp_path / fname)
        bamio.write_bam(p, refs, [tuple(r) for r in g["files"][fname]])
        paths.append(p)
    for mc, tag in ((-1, "0"), (20, "20"), (60, "60")):
        out = str(tmp_path / ("m%s.bin" % tag))
        r = ms.convert_files(paths, out, None, minimum_count=mc)
        assert open(out, "rb").read() == open(os.path.join(golden, "g4_multi_min%s.bin" % tag), "rb").read(), mc
        assert r["valid_alignments"] == g["counters"][str(mc)]["Number of alignments"]
