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


def _worker(rank, world, port, out_path, per_range=False):
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
    merged = ecdist.exchange_and_merge(eng, fresh(False), fresh(True), root=0, finalize_ranges=per_range)
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


@pytest.mark.parametrize("per_range", [False, True])
def test_two_processes_one_gpu_equal_one_handle(tmp_path, per_range):
    """Multisample over two ranks: the merged tables adopted by the root, or (``per_range``) every rank finalizing the key range it
    merged, the root assembling the rows and taking the ECs' hashes off them for the second exchange."""
    from alntools_amd import ecb, synth
    out = str(tmp_path / "merged.npz")
    mp.spawn(_worker, args=(2, _free_port(), out, per_range), nprocs=2, join=True)
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


@pytest.mark.parametrize("world", [2, 3])
def test_multisample_convert_over_several_ranks_writes_the_reference_bytes(tmp_path, monkeypatch, world):
    """``ALNTOOLS_GPUS=N`` for a directory of BAM files (``bam_utils_multisample.convert``): the files are dealt out to N processes
    (the reference: one worker per file, ``bam_utils_multisample.py:473-480``; here all on the one GPU of the box, over gloo), cell
    ids agreed on, EC tables merged by key range and adopted by rank 0, every rank's reads reduced to (EC, cell, file) triples
    against the merged ECs, rank 0 filters and writes: the reference's ``.bin`` bytes at its three thresholds, its range file,
    its counters -- three files on two ranks (uneven) and on three."""
    import json
    from alntools_amd import bam_utils_multisample as ms, bamio
    golden = os.path.join(os.path.dirname(__file__), "golden")
    monkeypatch.setenv("ALNTOOLS_GPUS", str(world))
    monkeypatch.setenv("ALNTOOLS_DIST_BACKEND", "gloo")
    monkeypatch.setenv("ALNTOOLS_GPU_LIST", ",".join(["0"] * world))
    g = json.load(open(os.path.join(golden, "g4_multi.json")))
    refs = [tuple(r) for r in g["references"]]
    paths = []
    for fname in g["glob_order"]:                      # the order the reference's glob returned when the golden was made
        p = str(tmp_path / fname)
        bamio.write_bam(p, refs, [tuple(r) for r in g["files"][fname]])
        paths.append(p)
    for mc, tag in ((-1, "0"), (20, "20"), (60, "60")):
        out, rng = str(tmp_path / ("m%s.bin" % tag)), str(tmp_path / ("m%s.range" % tag))
        r = ms.convert_files(paths, out, None, minimum_count=mc, range_filename=rng)
        assert open(out, "rb").read() == open(os.path.join(golden, "g4_multi_min%s.bin" % tag), "rb").read(), mc
        c = g["counters"][str(mc)]
        assert r["valid_alignments"] == c["Number of alignments"]
        assert r["n_ecs"] == c["Number of ECs after filtering"] and r["n_cells"] == c["Number of cells after filtering"]
        assert r["n_ecs_before"] == c["Number of ECs"] and r["n_cells_before"] == c["Number of cells"]
        assert open(rng).read() == open(os.path.join(golden, "g4_multi.range.txt")).read()


def _pieces_of(spec, world, dev):
    """The multi-rank protocol by hand in one process: ``world`` shard handles, cut into ``world`` key ranges, range q merged
    on its own handle and finalized there -> ([(packed, n_ecs, nnz)], totals)."""
    from alntools_amd import dist as ecdist
    from alntools_amd import ecb, synth
    R = spec.n_reads
    shards, base, bases, cuts, totals = [], 0, [], [], [0, 0, 0]
    for r in range(world):
        t = synth.generate(spec, r * R // world, (r + 1) * R // world, device=dev)
        b = ecb.EcBuilder(spec.n_loci, spec.n_haps, ec_capacity=1 << 12)
        b.push_device(t["read_id"], t["locus"], t["hapflag"])
        eng = ecdist.GpuEngine(b, dev)
        cuts.append(eng.table_export_parts(0, world))           # first reads from the shard's own 0, moved on by the receiver
        bases.append(base)
        a, v, n = eng.counters()
        totals = [totals[0] + a, totals[1] + v, totals[2] + n]
        base += n
        shards.append(b)
    pieces = []
    for q in range(world):
        part = ecdist.GpuEngine(ecb.EcBuilder(spec.n_loci, spec.n_haps, ec_capacity=1 << 12), dev)
        part.table_merge_many([(part.table_rebase(ent[eo[q] * 4:eo[q + 1] * 4], eo[q + 1] - eo[q], base), eo[q + 1] - eo[q], prs[po[q]:], po[q + 1] - po[q])
                               for (ent, prs, eo, po), base in zip(cuts, bases) if eo[q + 1] > eo[q]])
        pieces.append(part.finalize_range(*totals))
        part.b.close()
    for b in shards:
        b.close()
    return [p for p in pieces if p[1]], totals


@pytest.mark.parametrize("world", [1, 3, 8])
def test_ranges_finalized_apart_assemble_to_the_result_of_one_handle(world):
    """Finalize per key range (ecb_export_firsts_device / ecb_assemble_ranges_device): every range ranked and emitted on its
    own handle, the pieces placed by first read on an empty one == one handle over the whole stream, bit for bit."""
    from alntools_amd import dist as ecdist
    from alntools_amd import ecb, synth
    dev = torch.device("cuda:0")
    spec = synth.SynthSpec(80000, 3000, 8, paired=True)
    pieces, totals = _pieces_of(spec, world, dev)
    root = ecdist.GpuEngine(ecb.EcBuilder(spec.n_loci, spec.n_haps, ec_capacity=1 << 10), dev)
    s = root.assemble_ranges(pieces, *totals)
    assert root.b.finalize() == s                       # (idempotent: reports the assembled sizes)
    got = root.b.export()
    whole = synth.generate(spec, 0, spec.n_reads, device=dev)
    with ecb.EcBuilder(spec.n_loci, spec.n_haps) as one:
        one.push_device(whole["read_id"], whole["locus"], whole["hapflag"])
        exp_s = one.finalize()
        exp = one.export()
    assert s == exp_s
    for k in exp:
        assert np.array_equal(got[k], exp[k]), k
    with pytest.raises(ecb.EcbError):                   # no table behind an assembled result
        root.b.export_read_ec()
    root.b.reset()                                      # and the handle is as good as new
    root.b.push_device(whole["read_id"], whole["locus"], whole["hapflag"])
    assert root.b.finalize() == exp_s
    again = root.b.export()
    for k in exp:
        assert np.array_equal(again[k], exp[k]), k
    root.b.close()


def test_ranges_assemble_with_empty_shards_and_empty_ranges():
    """4 ranks, 3 reads: shards that hold nothing, key ranges that receive nothing."""
    from alntools_amd import dist as ecdist
    from alntools_amd import ecb, synth
    dev = torch.device("cuda:0")
    spec = synth.SynthSpec(3, 50, 2, paired=False)
    pieces, totals = _pieces_of(spec, 4, dev)
    assert 0 < len(pieces) <= 3
    root = ecdist.GpuEngine(ecb.EcBuilder(spec.n_loci, spec.n_haps, ec_capacity=1 << 10), dev)
    s = root.assemble_ranges(pieces, *totals)
    whole = synth.generate(spec, 0, spec.n_reads, device=dev)
    with ecb.EcBuilder(spec.n_loci, spec.n_haps) as one:
        one.push_device(whole["read_id"], whole["locus"], whole["hapflag"])
        assert one.finalize() == s
        exp = one.export()
    got = root.b.export()
    for k in exp:
        assert np.array_equal(got[k], exp[k]), k
    root.b.close()


def test_assembling_refuses_overlapping_or_malformed_pieces():
    from alntools_amd import dist as ecdist
    from alntools_amd import ecb, synth
    dev = torch.device("cuda:0")
    spec = synth.SynthSpec(20000, 500, 4, paired=False)
    pieces, totals = _pieces_of(spec, 2, dev)
    fresh = lambda: ecdist.GpuEngine(ecb.EcBuilder(spec.n_loci, spec.n_haps, ec_capacity=1 << 10), dev)
    e = fresh()
    with pytest.raises(ecb.EcbError) as ei:             # the same range twice: two ECs per first read
        e.assemble_ranges([pieces[0], pieces[0]], *totals)
    assert ei.value.code == -5
    e.b.close()
    e = fresh()
    with pytest.raises(ecb.EcbError) as ei:             # a first read the run does not have
        e.assemble_ranges(pieces, totals[0], totals[1], 10)
    assert ei.value.code == -5
    e.b.close()
    e = fresh()
    bad = pieces[1][0].clone()
    bad[1] = 1 << 30                                    # a row that ends beyond the piece
    with pytest.raises(ecb.EcbError) as ei:
        e.assemble_ranges([pieces[0], (bad, pieces[1][1], pieces[1][2])], *totals)
    assert ei.value.code == -5
    e.b.close()
    e = fresh()                                         # a handle that already holds reads does not assemble
    t = synth.generate(spec, 0, 100, device=dev)
    e.b.push_device(t["read_id"], t["locus"], t["hapflag"])
    with pytest.raises(ecb.EcbError) as ei:
        e.assemble_ranges(pieces, *totals)
    assert ei.value.code == -6
    e.b.close()
    e = ecdist.GpuEngine(ecb.EcBuilder(spec.n_loci, spec.n_haps, multisample=True), dev)      # a multisample root assembles too ...
    s_ms = e.assemble_ranges(pieces, *totals)
    assert s_ms["nnz_n"] == 0 and s_ms["n_samples"] == 0
    with pytest.raises(ecb.EcbError) as ei:             # ... and has no N until the shards' triples are adopted
        e.b.ms_filter(10, 1)
    assert ei.value.code == -6
    keys = torch.empty(s_ms["n_ecs"], dtype=torch.int64, device=dev)
    e.b.export_ec_keys_device(keys)                     # the ECs' hashes, off the assembled rows: the ones a table holds
    whole = synth.generate(spec, 0, spec.n_reads, device=dev)
    with ecb.EcBuilder(spec.n_loci, spec.n_haps) as one:
        one.push_device(whole["read_id"], whole["locus"], whole["hapflag"])
        one.finalize()
        exp_keys = torch.empty_like(keys)
        one.export_ec_keys_device(exp_keys)
    assert torch.equal(keys, exp_keys)
    e.b.close()
    # moving first reads on: not past 2^32 - 2 reads
    src = ecdist.GpuEngine(ecb.EcBuilder(spec.n_loci, spec.n_haps), dev)
    src.b.push_device(t["read_id"], t["locus"], t["hapflag"])
    ent, prs = src.table_export(0)
    ne, npairs, _ = src.table_sizes()
    e = fresh()
    with pytest.raises(ecb.EcbError) as ei:
        e.table_rebase(ent, ne, (1 << 32) - 1)
    assert ei.value.code == -8
    e.table_rebase(ent, ne, (1 << 32) - 20)                 # (queued; some first reads of the 100 are beyond 18)
    with pytest.raises(ecb.EcbError) as ei:
        e.table_merge(ent, ne, prs, npairs)
    assert ei.value.code == -5
    e.b.close(); src.b.close()


@pytest.mark.parametrize("world", [1, 2, 5])
def test_merge_inside_the_library_equals_one_handle(world):
    """``ecb_merge``: the multi-GPU merge for one process that drives several GPUs, inside libecb (peer copies instead of RCCL) --
    here with every shard on the one GPU of the box.  Contiguous read shards, one handle each -> the root == one handle over the
    whole stream, bit for bit; the spent shards and a root that is not empty are refused."""
    from alntools_amd import ecb, synth
    dev = torch.device("cuda:0")
    spec = synth.SynthSpec(90000, 3000, 8, paired=True)
    R = spec.n_reads
    shards = []
    for r in range(world):
        t = synth.generate(spec, r * R // world, (r + 1) * R // world, device=dev)
        b = ecb.EcBuilder(spec.n_loci, spec.n_haps, ec_capacity=1 << 12)
        b.push_device(t["read_id"], t["locus"], t["hapflag"])
        shards.append(b)
    root = ecb.EcBuilder(spec.n_loci, spec.n_haps, ec_capacity=1 << 10)
    s = root.merge_from(shards)
    got = root.export()
    whole = synth.generate(spec, 0, R, device=dev)
    with ecb.EcBuilder(spec.n_loci, spec.n_haps) as one:
        one.push_device(whole["read_id"], whole["locus"], whole["hapflag"])
        exp_s = one.finalize()
        exp = one.export()
    assert s == exp_s and root.finalize() == exp_s
    for k in exp:
        assert np.array_equal(got[k], exp[k]), k
    with pytest.raises(ecb.EcbError):                   # the root holds a result now
        root.merge_from(shards)
    for b in shards:
        b.close()
    root.close()
