"""The multi-GPU protocol of alntools_amd/dist.py with real libecb handles in two processes sharing the one GPU of the
test box.  RCCL refuses two ranks on one device, so the process group is gloo and dist.HostStagedEngine stages the protocol's
tensors through host memory; everything else (C ABI calls, kernels, message pattern) is what runs over RCCL on 8 GPUs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _meta(g):
    g = g.astype(np.uint64)
    return (((g * np.uint64(2654435761)) % np.uint64(101)) | ((g % np.uint64(3)) << np.uint64(22))).astype(np.uint32)


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from alntools_amd import dist as ecdist
    from alntools_amd import ecb, synth
    dev = torch.device("cuda:0")
    spec = synth.SynthSpec(60000, 3000, 8, paired=True)
    R = spec.n_reads
    a, b_ = rank * R // world, (rank + 1) * R // world
    base = synth.generate(spec, 0, a)["n_reads"] if a else 0
    t = synth.generate(spec, a, b_, device=dev)
    b = ecb.EcBuilder(spec.n_loci, spec.n_haps, ec_capacity=1 << 12, multisample=True)
    b.push_device(t["read_id"], t["locus"], t["hapflag"])
    b.push_cells(_meta(np.arange(base, base + t["n_reads"])), 0)
    eng = ecdist.HostStagedEngine(ecdist.GpuEngine(b, dev))
    fresh = lambda ms: (lambda: ecdist.HostStagedEngine(ecdist.GpuEngine(ecb.EcBuilder(spec.n_loci, spec.n_haps, ec_capacity=1 << 12, multisample=ms), dev)))
    merged = ecdist.exchange_and_merge(eng, fresh(False), fresh(True), root=0)
    n_ecs = None
    if rank == 0:
        s = merged.e.b.finalize()
        n_ecs = s["n_ecs"]
    nt = ecdist.exchange_multisample(eng, merged, n_ecs, root=0)
    if rank == 0:
        pr = merged.e.b.export_pairs()
        out = merged.e.b.export()
        np.savez(out_path, nt=nt, n_ecs=n_ecs, n_reads=s["n_reads"], all=s["all_alignments"], valid=s["valid_alignments"],
                 **{"p_" + k: v for k, v in pr.items()}, **out)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_processes_one_gpu_equal_one_handle(tmp_path):
    from alntools_amd import ecb, synth
    out = str(tmp_path / "merged.npz")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    spec = synth.SynthSpec(60000, 3000, 8, paired=True)
    whole = synth.generate(spec, 0, spec.n_reads, device=torch.device("cuda:0"))
    with ecb.EcBuilder(spec.n_loci, spec.n_haps, multisample=True) as one:
        one.push_device(whole["read_id"], whole["locus"], whole["hapflag"])
        one.push_cells(_meta(np.arange(whole["n_reads"])), 0)
        s = one.finalize()
        exp_p, exp_a = one.export_pairs(), one.export()
    assert int(got["n_ecs"]) == s["n_ecs"] and int(got["nt"]) == s["nnz_n"] and int(got["n_reads"]) == s["n_reads"]
    assert int(got["all"]) == s["all_alignments"] and int(got["valid"]) == s["valid_alignments"]
    for k in ("ec", "cell", "file", "count", "first"):
        assert np.array_equal(got["p_" + k], exp_p[k]), k
    for k in ("indptrA", "indicesA", "dataA"):
        assert np.array_equal(got[k], exp_a[k]), k


@pytest.mark.parametrize("world", [2, 3])
def test_convert_over_several_ranks_writes_the_reference_bytes(tmp_path, monkeypatch, world):
    """``ALNTOOLS_GPUS=N``: convert() itself runs the multi-rank path -- N processes (here all on the one GPU of the box, over
    gloo with host-staged tables), contiguous read shards of the decoded BAM, merge by key range, rank 0 writes.  The .bin and
    the range file must be the bytes the reference wrote for the same BAM (edge-case and config-1 goldens)."""
    import json
    from alntools_amd import bam_utils, bamio
    golden = os.path.join(os.path.dirname(__file__), "golden")
    monkeypatch.setenv("ALNTOOLS_GPUS", str(world))
    monkeypatch.setenv("ALNTOOLS_DIST_BACKEND", "gloo")
    monkeypatch.setenv("ALNTOOLS_GPU_LIST", ",".join(["0"] * world))
    for name in ("g1_edge", "g2_c1"):
        g = json.load(open(os.path.join(golden, name + ".json")))
        bam = str(tmp_path / g["sample"])
        if "records" in g:
            bamio.write_bam(bam, [tuple(r) for r in g["references"]], [tuple(r) for r in g["records"]])
            targets = None
        else:
            from alntools_amd import synth
            spec = synth.SynthSpec(**g["spec"])
            bamio.write_bam(bam, spec.references(), synth.raw_records(spec, 0, spec.n_reads), level=1)
            targets = None
        out, rng = str(tmp_path / (name + ".bin")), str(tmp_path / (name + ".range"))
        sizes = bam_utils.convert(bam, out, None, range_filename=rng, target_filename=targets)
        assert open(out, "rb").read() == open(os.path.join(golden, name + ".bin"), "rb").read(), name
        exp_rng = os.path.join(golden, name + ".range.txt")
        if os.path.exists(exp_rng):
            assert open(rng).read() == open(exp_rng).read(), name
        assert sizes["n_ecs"] > 0
