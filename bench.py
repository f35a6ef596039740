#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""bench.py -- alignments/sec BAM->EC on MI355X (BASELINE.json metric).

One "step" = one whole pass of the hot path over the workload's record tuples, already
resident in HBM: reset the EC table, stream every record through libecb's k_stream kernel
(filter, read segmentation, per-read target sets, EC lookup/insert), count reads per EC (k_count), exchange + merge
the per-GPU EC tables when N > 1 (by key range, RCCL point-to-point), rank ECs by first appearance and emit CSR A / N.

Workload (default): BASELINE config 3 -- 100 M paired-end reads, 8 haplotypes x 80 k
transcripts (~3.3 G BAM records, ~40 GB of tuples), synthetic (alntools_amd/synth.py, seed
20260101), read-sharded contiguously over the N GPUs (strong scaling: total work fixed).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c1|tiny]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (reads, loci, haplotypes, paired, description)
    "c3": (100_000_000, 80_000, 8, True, "BASELINE config 3: 100M paired-end reads, 8 hap x 80k transcripts, bam2emase path"),
    "c2": (50_000_000, 40_000, 8, False, "BASELINE config 2: 50M single-end reads, 8 hap x 40k transcripts, bam2ec path"),
    "c1": (10_000, 1_000, 2, False, "BASELINE config 1: 10k single-end reads, 2 hap x 1k transcripts"),
    "tiny": (400_000, 4_000, 8, True, "smoke-sized paired-end workload"),
    "c3h": (50_000_000, 80_000, 8, True, "half of config 3 (debug)"),
    "c2x": (150_000_000, 40_000, 8, False, "3x config 2 (debug: > 2^31 records)"),
    "c4": (200_000_000, 80_000, 8, True, "BASELINE config 4: multisample, 200M paired-end reads, 5k cell barcodes (log-normal sizes) over 64 files, minimum count 1000"),
    "c4h": (100_000_000, 80_000, 8, True, "half of config 4 (debug)"),
    "c4t": (2_000_000, 8_000, 8, True, "config 4's shape at 2 M reads (debug: rehearsals of the sharded multisample step)"),
    "c3s": (100_000_000, 80_000, 8, True, "config 3 with every read's loci 64 target ids apart (their low bits equal: the worst case for the stream kernel's LDS table)"),
    "c3q": (100_000_000, 80_000, 8, True, "config 3 with every read's loci 4 target ids apart: the third locus of a read collides with its first in the LDS table (reads of three loci and more: two in five)"),
    "c3r": (100_000_000, 80_000, 8, True, "config 3 with paralogs: 30 % of the reads hit 1 - 3 further loci drawn uniformly over all targets besides their cluster of consecutive ids (multi-mappers at unrelated target ids: first-probe collisions in the LDS table at the birthday rate)"),
    "dip": (200_000_000, 40_000, 2, False, "diploid single-end reads: 2 hap x 40k transcripts, ~4 records per read (short reads: several passes per tile)"),
}
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)


def workload_spec(name):
    """The synthetic stream a workload name stands for."""
    from alntools_amd import synth
    R, T, H, paired, _ = WORKLOADS[name]
    return synth.SynthSpec(R, T, H, paired=paired, locus_stride={"c3s": 64, "c3q": 4}.get(name, 1), paralog_pct=30 if name == "c3r" else 0)


def generate_shard(spec, r0, r1, device, chunk_reads=1 << 20):
    """Tuples of reads [r0, r1) as three int32 CUDA tensors (uint32 bit patterns); read ids local from 0."""
    import torch
    from alntools_amd import synth
    n = synth.count_records(spec, r0, r1, device=device)
    # ONE allocation for the three streams (each starting on a 2 MiB boundary within it).  Three allocations of 13 GB each land wherever the
    # driver has room, and where they land moves a pure read sweep over them by 4.5 % and k_stream by 8 - 10 % (profiles/r04_stream_placement*.txt:
    # the pool's "slow" and "fast" boxes of round 3 were slow and fast RUNS); one allocation of all three measured at the fast end in every
    # arrangement tried.  What a caller of ecb_push_device should do, too (include/ecb.h).
    stride = (n * 4 + (2 << 20) - 1) // (2 << 20) * (2 << 20) // 4
    arena = torch.empty(3 * stride, dtype=torch.int32, device=device)
    rid, loc, hf = arena[:n], arena[stride:stride + n], arena[2 * stride:2 * stride + n]
    at, reads, valid = 0, 0, 0
    for a in range(r0, r1, chunk_reads):
        g = synth.generate(spec, a, min(a + chunk_reads, r1), device=device, read_id_base=reads)
        m = g["n_records"]
        rid[at:at + m] = g["read_id"]
        loc[at:at + m] = g["locus"]
        hf[at:at + m] = g["hapflag"]
        at += m
        reads += g["n_reads"]
        valid += g["n_valid"]
        del g
    assert at == n
    return rid, loc, hf, dict(records=n, reads=reads, valid=valid)


def generate_tiles(spec, r0, r1, device, chunk_reads=1 << 20):
    """The same tuples in the layout of ``ecb_push_device_tiled``: ONE int32 tensor of whole tiles, tile t = words [1536 t, 1536 t + 1536) = the 512
    read ids | 512 loci | 512 haplotype/flag words of records [512 t, 512 t + 512).  -> (tiles, counts as generate_shard's)."""
    import torch
    from alntools_amd import synth
    n = synth.count_records(spec, r0, r1, device=device)
    nt = (n + 511) // 512
    tiles = torch.zeros(nt * 1536, dtype=torch.int32, device=device)
    cols = [torch.as_strided(tiles, (nt, 512), (1536, 1), 512 * k) for k in range(3)]      # stream k as rows of 512 records

    def put(col, at, src):                       # src -> records [at, at + len(src)) of a stream
        m = src.numel()
        head = min(m, (512 - at % 512) % 512)
        if head:
            col[at // 512, at % 512:at % 512 + head] = src[:head]
        rows = (m - head) // 512
        if rows:
            col[(at + head) // 512:(at + head) // 512 + rows] = src[head:head + rows * 512].view(rows, 512)
        tail = m - head - rows * 512
        if tail:
            col[(at + m) // 512, :tail] = src[m - tail:]

    at, reads, valid = 0, 0, 0
    for a in range(r0, r1, chunk_reads):
        g = synth.generate(spec, a, min(a + chunk_reads, r1), device=device, read_id_base=reads)
        for col, key in zip(cols, ("read_id", "locus", "hapflag")):
            put(col, at, g[key])
        at += g["n_records"]
        reads += g["n_reads"]
        valid += g["n_valid"]
        del g
    assert at == n
    return tiles, dict(records=n, reads=reads, valid=valid)


def tiles_to_arrays(tiles, n):
    """Three contiguous int32 tensors out of a tensor of whole tiles (copies: the oracle legs and the host-path legs want arrays)."""
    import torch
    nt = (n + 511) // 512
    return tuple(torch.as_strided(tiles, (nt, 512), (1536, 1), 512 * k).reshape(-1)[:n].contiguous() for k in range(3))


def _host_slice(rid, loc, hf, n_reads):
    """The first ``n_reads`` reads of the device-resident stream as host uint32 arrays (whole reads)."""
    import numpy as np
    import torch
    cut = int(torch.searchsorted(rid, torch.tensor([n_reads], dtype=torch.int32, device=rid.device))[0]) if n_reads else 0
    if cut == 0 or n_reads >= int(rid[-1]) + 1:
        cut = rid.numel()
    return [t[:cut].cpu().numpy().view(np.uint32) for t in (rid, loc, hf)], cut


def cpu_baseline(rid, loc, hf, n_haps, sample_reads, threads=0):
    """The oracle's C restatement (oracle/ec_oracle.c) on the host's hardware threads: all of them unless ``threads`` says
    otherwise, contiguous read shards + ordered merge.  -> (the bench line's object, the oracle's result when the sample was
    the whole workload -- what the GPU's CSR is then compared with, bit for bit -- else None)."""
    from oracle import c_oracle
    h, cut = _host_slice(rid, loc, hf, sample_reads)
    cores = threads or os.cpu_count() or 1
    c_oracle.load()
    t0 = time.perf_counter()
    r = c_oracle.ec_from_tuples(h[0], h[1], h[2], n_haps, threads=cores)
    dt = time.perf_counter() - t0
    whole = cut == rid.numel()
    line = dict(value=cut / dt, unit="alignments/s", cores=cores, kind="port",
                sample="%s (%d reads, %d records), %d ECs; oracle/ec_oracle.c, %d threads over contiguous read shards + "
                       "ordered merge; %.2f s" % ("the whole workload" if whole else "first reads of the same stream",
                                                   r["n_reads"], cut, len(r["count"]), cores, dt))
    return line, (r if whole else None)


def parity_vs_oracle(b, push, exp):
    """The GPU's result on the whole workload against the C oracle's on the same tuples: CSR A (indptr, indices, data), the
    counts and the three counters, bit for bit.  Outside the timed region.  -> (equal?, what differs)."""
    import numpy as np
    b.reset()
    push()                                     # (the product path as the timed steps took it: tiles or arrays)
    sizes = b.finalize()
    out = b.export()
    diff = [k for k, (a, e) in dict(indptr=(out["indptrA"], exp["indptr"]), indices=(out["indicesA"], exp["indices"]),
                                    data=(out["dataA"], exp["data"]), count=(out["dataN"], exp["count"])).items()
            if not np.array_equal(np.asarray(a, dtype=np.int64), np.asarray(e, dtype=np.int64))]
    diff += [k for k, e in (("all_alignments", exp["n_all"]), ("valid_alignments", exp["n_valid"]), ("n_reads", exp["n_reads"]))
             if int(sizes[k]) != int(e)]
    return not diff, diff


def cpu_baseline_py(rid, loc, hf, n_haps, sample_reads):
    """The reference's own algorithm in Python (oracle/py_baseline.py: per-alignment loop, string keys, ordered dicts, ordered
    merge, per-EC incidence build) on a slice, P = os.cpu_count() worker processes; per-alignment cost is O(1), so the
    full-workload time is the slice's time scaled by records."""
    from oracle import py_baseline
    h, cut = _host_slice(rid, loc, hf, sample_reads)
    P = os.cpu_count() or 1
    r = py_baseline.run(h[0], h[1], h[2], n_haps, P)      # (workers are spawned, not forked: this process holds a HIP context)
    dt = r["seconds_scan"] + r["seconds_merge"] + r["seconds_build"]
    return dict(value=cut / dt, unit="alignments/s", cores=P, kind="port (Python restatement of the reference's loops)",
                sample="first %d reads (%d records) of the same stream, %d ECs; %d worker processes over contiguous read shards; "
                       "scan %.2f s + ordered merge %.2f s + A build %.2f s; scales linearly in records (extrapolated whole-"
                       "workload time: %.0f s)" % (sample_reads, cut, r["n_ecs"], P, r["seconds_scan"], r["seconds_merge"],
                                                   r["seconds_build"], dt * rid.numel() / cut))


def h2d_inclusive(b, rid, loc, hf, n_records):
    """Records/s through the host-pointer entry point (ecb_push: staging, H2D copies, kernels), from pinned host memory."""
    import numpy as np
    import torch
    m = min(n_records, rid.numel())
    last = int(rid[m - 1])
    cut = int(torch.searchsorted(rid, torch.tensor([last], dtype=torch.int32, device=rid.device))[0]) or m    # whole reads
    host = [torch.empty(cut, dtype=torch.int32).pin_memory() for _ in range(3)]
    for d, t in zip(host, (rid, loc, hf)):
        d.copy_(t[:cut])
    torch.cuda.synchronize()
    arrs = [d.numpy().view(np.uint32) for d in host]
    b.reset()
    t0 = time.perf_counter()
    b.push(*arrs)
    b.finalize()
    dt = time.perf_counter() - t0
    return dict(value=cut / dt, unit="alignments/s", records=cut, seconds=dt,
                note="pinned host arrays -> ecb_push (16 Mi-record staging batches, H2D copies, kernels) -> finalize; PCIe Gen5 x16 "
                     "moves at most 63 GB/s = 5.2 G records/s of 12-byte tuples")


def e2e_from_bam(spec_args, reads, tmpdir):
    """BAM file -> .bin through the drop-in's convert() (host decode + tuple encoding + device + writer), decode time apart."""
    from alntools_amd import bam_utils, bamio, synth
    from alntools_amd.tuples import HeaderMaps, TupleEncoder
    spec = synth.SynthSpec(*spec_args[:3], paired=spec_args[3])
    bam, out = os.path.join(tmpdir, "e2e.bam"), os.path.join(tmpdir, "e2e.bin")
    bamio.write_bam(bam, spec.references(), synth.raw_records(spec, 0, reads), level=1)
    t0 = time.perf_counter()
    sizes = bam_utils.convert(bam, out, None)
    t_all = time.perf_counter() - t0
    def host_side(names):                             # the host side alone: decode + tuple encoding, nothing pushed
        t0 = time.perf_counter()
        rd = bam_utils.open_bam(bam, names=names)
        enc = TupleEncoder(HeaderMaps(rd.references, rd.lengths))
        t_head = time.perf_counter() - t0             # the @SQ lines: a fixed cost per file (320 k names at config 2), not per record
        n = sum(len(t["read_id"]) for t in bam_utils.iter_tuple_batches(rd, enc))
        rd.close()
        return n, time.perf_counter() - t0, type(rd).__name__, t_head
    n, t_host, decoder, t_head = host_side(False)     # what convert() uses: the native decoder (csrc/bamdec.c) when it is built
    _, t_py, py_decoder, _ = host_side(True)          # the pure-Python reader of the same file (pysam's stand-in), for scale
    return dict(value=n / t_all, unit="alignments/s", records=n, reads=reads, seconds=t_all, host_decode_encode_seconds=t_host,
                host_header_seconds=t_head, host_records_per_second_after_header=n / max(t_host - t_head, 1e-9),
                decoder=decoder, host_decode_encode_seconds_python_reader=t_py, python_reader=py_decoder, ecs=sizes["n_ecs"],
                note="convert(bam, bin): BAM decode on the host (%s), tuple encoding, ecb_push, finalize, .bin writer; header maps of "
                     "all @SQ names included (host_header_seconds: a fixed cost per file, most of a slice this small)" % decoder)


def secondary(name, device, local, steps, with_oracle, layout="arrays"):
    """A further workload behind the headline's timed steps, on the same box in the same run: ``steps`` whole steps (reset -> push ->
    finalize) after one warm-up, the stream kernel's own HIP-event time, and the result held to the C oracle's on the whole workload
    (workloads above a billion records: the device's own exactness pass instead -- every read's target set against its EC's key)."""
    import torch
    from alntools_amd import ecb
    R, T, H, paired, desc = WORKLOADS[name]
    spec = workload_spec(name)
    tiled = layout == "tiles"
    if tiled:
        tiles, st = generate_tiles(spec, 0, R, device)
        rid = loc = hf = None
    else:
        rid, loc, hf, st = generate_shard(spec, 0, R, device)
    n_rec = st["records"]

    def push():
        if tiled:
            b.push_device_tiled(tiles, n_rec)
        else:
            b.push_device(rid, loc, hf)
    cap = 1 << (25 if name == "c3r" else 24 if name.startswith("c3") else 22)
    out = {"workload": desc, "records": st["records"], "reads_with_alignments": st["reads"], "layout": layout}
    with ecb.EcBuilder(T, H, device=local, ec_capacity=cap, arena_capacity=1 << 26) as b:
        b.hint_reads(st["reads"])
        sizes = {}

        def step():
            b.reset()
            push()
            sizes.update(b.finalize())

        step()
        b.profile(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        k_ms, k_launches, _ = b.profile_read()
        kernel = b.profile_kernel()
        b.profile(False)
        alg = 12.0 * st["records"] + 4.0 * st["reads"]
        k_per = k_ms / max(steps, 1)
        out.update(ms_per_step=dt * 1e3 / steps, k_stream_ms=k_per, launches_per_step=k_launches / float(max(steps, 1)), kernel=kernel,
                   achieved=alg / (k_per * 1e-3) / 1e9, frac=alg / (k_per * 1e-3) / 1e9 / HBM_PEAK_GBS, algorithmic_bytes_per_launch=alg,
                   value=st["records"] / (dt / steps), ecs=sizes.get("n_ecs"), nnz_a=sizes.get("nnz_a"))
        if with_oracle and st["records"] <= 1_000_000_000:
            base, oracle_result = cpu_baseline(*(tiles_to_arrays(tiles, n_rec) if tiled else (rid, loc, hf)), H, 0)
            ok, diff = parity_vs_oracle(b, push, oracle_result)
            out["parity_vs_oracle"] = ok
            out["cpu_baseline"] = {"value": base["value"], "cores": base["cores"], "kind": base["kind"]}
            if not ok:
                out["differs"] = diff
            del oracle_result
        else:
            b.reset()
            push()
            bad, skipped = b.verify_device_tiled(tiles, n_rec) if tiled else b.verify_device(rid, loc, hf)
            out["parity_vs_oracle"] = None
            out["exactness_pass"] = {"reads_differing_from_their_ec_key": bad, "reads_on_the_long_read_path": skipped}
    del rid, loc, hf
    if tiled:
        del tiles
    torch.cuda.empty_cache()
    return out


def placement_spread(b, tensors, step_from, n_arenas=3, steps=3):
    """k_stream's time with the SAME tuples at other places in HBM: ``n_arenas`` further allocations of the tuples' size, the tuples copied
    into each, ``steps`` steps from each (untimed for the headline, which was measured where the generator put the tuples).  ``tensors``: the
    one tensor of whole tiles, or the three arrays (which then share ONE further allocation each time, as generate_shard lays them out).
    With three arrays the time moves by up to 14 % with where the 40 GB sit, reproducibly per allocation (profiles/r04_stream_regions*.txt);
    with whole tiles by somewhat less (profiles/r04_tiles_against_arrays.txt).  -> ms per launch, the run's own allocation first."""
    import torch
    out, arenas = [], []
    try:
        for i in range(n_arenas + 1):
            if i == 0:
                views = tensors
            elif len(tensors) == 1:
                a = torch.empty_like(tensors[0])
                arenas.append(a)              # (all kept until the end: allocations that exist at the same time are different memory)
                a.copy_(tensors[0])
                views = (a,)
            else:
                n = tensors[0].numel()
                stride = (n * 4 + (2 << 20) - 1) // (2 << 20) * (2 << 20) // 4
                a = torch.empty(3 * stride, dtype=torch.int32, device=tensors[0].device)
                arenas.append(a)
                views = (a[:n], a[stride:stride + n], a[2 * stride:2 * stride + n])
                for v, src in zip(views, tensors):
                    v.copy_(src)
            torch.cuda.synchronize()
            step_from(*views)
            b.profile(True)
            for _ in range(steps):
                step_from(*views)
            torch.cuda.synchronize()
            ms, launches, _ = b.profile_read()
            b.profile(False)
            out.append(round(ms / max(launches, 1), 3))
    except RuntimeError:                      # (out of memory for another copy: report what there is)
        pass
    del arenas
    torch.cuda.empty_cache()
    return out


def copy_peak(device, nbytes=4 << 30, reps=5):
    """Device-to-device copy rate (read + written bytes per second) measured in this run with the access shape of the stream
    kernel -- 16 bytes per lane, non-temporal (tools/micro/copy_peak.hip) -- and the rate of reading alone: what a kernel can
    get out of this card's HBM.  -> (copy GB/s, read GB/s, how it was measured)."""
    import ctypes
    import torch
    a = torch.empty(nbytes // 4, dtype=torch.int32, device=device).fill_(1)
    c = torch.empty_like(a)
    lib = os.path.join(ROOT, "tools", "micro", "bin", "libcopy_peak.so")
    if os.path.exists(lib):
        L = ctypes.CDLL(lib)
        for f in (L.copy_peak_run, L.read_peak_run):
            f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
        torch.cuda.synchronize()
        g, r = ctypes.c_double(0), ctypes.c_double(0)
        if L.copy_peak_run(a.data_ptr(), c.data_ptr(), nbytes, reps, ctypes.byref(g)) == 0 and \
                L.read_peak_run(a.data_ptr(), c.data_ptr(), nbytes, reps, ctypes.byref(r)) == 0:
            return g.value, r.value, "tools/micro/copy_peak.hip: 16 B per lane, non-temporal, %d x %d MiB" % (reps, nbytes >> 20)
    c.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        c.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9, None, "torch.Tensor.copy_ (tools/micro/bin/libcopy_peak.so not built)"


def sensors_under_load(step, fence, device_index, reps=8):
    """What the card's own sensors say while ``step`` runs (``reps`` untimed steps after the timed ones; a sampling thread reads
    the amdgpu hwmon files every half millisecond): shader and memory clock, socket power against its cap, temperatures.  Boxes of
    one pool move the same machine code at different speeds; this is the context a time was measured in, not part of the metric.
    -> dict, or None where the files are not there to read."""
    import glob
    import threading
    import torch
    want = None
    try:
        pr = torch.cuda.get_device_properties(device_index)
        want = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
    except Exception:
        pass
    cands = []
    for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        if os.path.exists(os.path.join(hw, "freq1_input")):
            pci = os.path.basename(os.path.realpath(os.path.join(hw, "..", "..")))
            cands.append((0 if (want and pci == want) else 1, hw))
    if not cands:
        return None
    hw = sorted(cands)[0][1]

    def rd(name):
        try:
            with open(os.path.join(hw, name)) as f:
                return float(f.read().strip())
        except (OSError, ValueError):
            return None
    temps = {}
    for t in sorted(glob.glob(os.path.join(hw, "temp*_input"))):
        lab = t.replace("_input", "_label")
        temps[os.path.basename(t)] = open(lab).read().strip() if os.path.exists(lab) else os.path.basename(t)[:-6]
    rows, stop = [], threading.Event()

    def sample():
        while not stop.is_set():
            rows.append((rd("freq1_input"), rd("freq2_input"), rd("power1_input")) + tuple(rd(k) for k in temps))
            time.sleep(0.0005)
    fence()
    th = threading.Thread(target=sample, daemon=True)
    th.start()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    fence()
    dt = time.perf_counter() - t0
    stop.set()
    th.join()

    def stat(col, scale):
        v = [r[col] * scale for r in rows if r[col] is not None]
        return {"avg": round(sum(v) / len(v), 1), "min": round(min(v), 1), "max": round(max(v), 1)} if v else None
    cap = rd("power1_cap")
    return {"sclk_mhz": stat(0, 1e-6), "mclk_mhz": stat(1, 1e-6), "socket_power_w": stat(2, 1e-6),
            "power_cap_w": round(cap * 1e-6, 1) if cap else None,
            "temperature_c": {lab: stat(3 + i, 1e-3) for i, lab in enumerate(temps.values())},
            "samples": len(rows), "over": "%d untimed steps, %.1f ms" % (reps, dt * 1e3), "source": hw}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-sample-reads", type=int, default=0, help="reads of the C baseline's sample (0 = the whole workload)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the C baseline (0 = every hardware thread of the host)")
    ap.add_argument("--py-sample-reads", type=int, default=1_000_000, help="reads of the Python restatement's slice")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip every host-side measurement (C / Python baselines, H2D, BAM)")
    ap.add_argument("--layout", default="arrays", choices=("arrays", "tiles"),
                    help="how the resident tuples sit in HBM: three arrays in one allocation (ecb_push_device; the default: what the library's own staging for "
                         "host batches uses) or one buffer of whole tiles (ecb_push_device_tiled)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary workloads (c2, c3r, dip) behind a default config-3 run")
    ap.add_argument("--secondary-steps", type=int, default=5)
    ap.add_argument("--e2e-slice-reads", type=int, default=100_000, help="reads of the config-2 slice converted from a real BAM file")
    args = ap.parse_args()

    # (before anything can touch the GPU runtime: the host driver of this pool only supports dmabuf IPC, and the variable is read
    #  when the runtime starts, not when the process group does)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from alntools_amd import dist as ecdist
    from alntools_amd import ecb, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d" %
                             (args.gpus, args.gpus))
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libecb has no CPU path")
    # ECB_DIST_BACKEND=gloo: functional rehearsal of the N > 1 code path with all ranks on ONE GPU (RCCL refuses that); tables
    # are staged through host memory, so its numbers mean nothing
    rehearsal = os.environ.get("ECB_DIST_BACKEND", "nccl") == "gloo"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    use_dist = world > 1 or bool(os.environ.get("ECB_FORCE_DIST"))   # ECB_FORCE_DIST: exercise the RCCL path at N = 1
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    red_dev = torch.device("cpu") if rehearsal else device          # where the small reductions of the timing live

    R, T, H, paired, desc = WORKLOADS[args.workload]
    spec = workload_spec(args.workload)
    r0, r1 = rank * R // world, (rank + 1) * R // world
    t_gen = time.perf_counter()
    tiled = args.layout == "tiles"
    tiles = rid = loc = hf = None
    if tiled:
        tiles, st = generate_tiles(spec, r0, r1, device)
    else:
        rid, loc, hf, st = generate_shard(spec, r0, r1, device)
    n_rec = st["records"]
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen

    def push():                                   # the resident tuples into the handle: whole tiles or three arrays
        if tiled:
            b.push_device_tiled(tiles, n_rec)
        else:
            b.push_device(rid, loc, hf)

    # EC-table slots: sized for the workload's EC count (c3: 3.7 M ECs) so that the timed steps do not grow it
    # (a shard of 1/4 or 1/8 of config 3 still founds 2.3 - 3 M of its 3.7 M ECs: 2^23 slots keep it under half full)
    multisample = args.workload.startswith("c4")
    ec_cap = 1 << int(os.environ.get("ECB_EC_CAP_LOG2", "25" if args.workload == "c3r" else ("23" if world >= 4 else "24") if args.workload in ("c3", "c3h", "c3s", "c3q", "c4", "c4h") else "22"))
    b = ecb.EcBuilder(T, H, device=local, ec_capacity=ec_cap, arena_capacity=1 << 26, multisample=multisample)
    hint = 0 if os.environ.get("ECB_NO_HINT") else st["reads"]      # (a caller that knows how many reads its stream holds says so: one host wait per push)
    b.hint_reads(hint)
    eng = ecdist.GpuEngine(b, device)
    if rehearsal:
        eng = ecdist.HostStagedEngine(eng)
    root_eng = part_eng = None
    if use_dist:        # the key range this rank merges (1/world of the ECs), and on rank 0 the table that adopts all ranges
        # (the merging table is sized for the entries that ARRIVE -- 1/world of every shard's ECs, i.e. about one shard's worth, all of which
        #  the merge must be able to seat before it knows how many of them are the same EC -- so that no timed step grows it; a table that
        #  is merged into keeps a list of its occupied slots, so its size costs nothing per step.  The root's handle only assembles.)
        part_cap = ec_cap
        part_eng = ecdist.GpuEngine(ecb.EcBuilder(T, H, device=local, ec_capacity=part_cap, arena_capacity=1 << 26), device)
        if rank == 0:
            root_eng = ecdist.GpuEngine(ecb.EcBuilder(T, H, device=local, ec_capacity=(1 << 22) if args.workload in ("c3", "c3h", "c4", "c4h") else 1 << 20,
                                                           arena_capacity=1 << 26, multisample=multisample), device)
        if rehearsal:
            part_eng = ecdist.HostStagedEngine(part_eng)
            root_eng = ecdist.HostStagedEngine(root_eng) if root_eng is not None else None

    def make_root():
        root_eng.b.reset()
        return root_eng

    def make_part():
        part_eng.b.reset()
        return part_eng

    sizes = {}
    ms_sizes = {}
    parity_failed = False
    n_cells, min_count, meta = 5000, (100 if args.workload == "c4t" else 1000), None
    if multisample:
        # cell of every read: 5 000 barcodes with log-normal sizes (sigma = 1), drawn by a hash of the read's index in the RUN; file = the
        # read's 64th of the run (a directory of 64 BAM files read in order, dealt out to the ranks in contiguous runs as
        # bam_utils_multisample.convert does); cell ids compacted over the cells the run uses (the host names cells as they turn up)
        first_read, run_reads = 0, st["reads"]
        if use_dist:
            mine_n = torch.tensor([st["reads"]], dtype=torch.int64, device=red_dev)
            all_n = torch.empty(world, dtype=torch.int64, device=red_dev)
            dist.all_gather_into_tensor(all_n, mine_n)
            all_n = [int(x) for x in all_n.cpu().tolist()]
            first_read, run_reads = sum(all_n[:rank]), sum(all_n)
        g = torch.arange(first_read, first_read + st["reads"], dtype=torch.int64, device=device)
        u = (((g * 0x9E3779B97F4A7C15) >> 11) & ((1 << 40) - 1)).to(torch.float64) / float(1 << 40)
        w = torch.exp(torch.randn(n_cells, generator=torch.Generator().manual_seed(20260101), dtype=torch.float64)).to(device)
        cdf = torch.cumsum(w / w.sum(), 0)
        cell = torch.searchsorted(cdf, u).clamp_(max=n_cells - 1)
        used = torch.zeros(n_cells, dtype=torch.int64, device=device)
        used[cell] = 1
        if use_dist:
            used_r = used.to(red_dev)
            dist.all_reduce(used_r, op=dist.ReduceOp.MAX)
            used = used_r.to(device)
        compact = torch.cumsum(used, 0) - 1
        meta = (compact[cell] | (((g * 64) // run_reads) << 22)).to(torch.int32)
        del g, u, cell, used, compact
    per_range = os.environ.get("ECB_DIST_FINALIZE", "ranges") != "root"      # ("root": the merged tables go to rank 0, which finalizes alone)

    def step():
        b.reset()
        push()
        if multisample and use_dist:      # config 4 as BASELINE names it: per-barcode EC build over the GPUs
            b.push_cells_device(meta, 0)
            m = ecdist.exchange_and_merge(eng, make_part, make_root, root=0, finalize_ranges=per_range)
            n_ecs = None
            if m is not None:
                sizes.update(m.b.finalize())
                n_ecs = sizes["n_ecs"]
            nt = ecdist.exchange_multisample(eng, m, n_ecs, root=0)      # every rank's reads -> (EC, cell, file) triples against the merged ECs
            if m is not None:
                sizes["nnz_n"] = nt
            if m is not None:
                ms_sizes.update(m.b.ms_filter_sizes(n_cells, min_count))
        elif multisample:      # + the cell of every read (4 B per read), the (EC, cell, file) triples, cell order / filter / N on the device
            b.push_cells_device(meta, 0)
            sizes.update(b.finalize())
            ms_sizes.update(b.ms_filter_sizes(n_cells, min_count))
        elif use_dist:
            m = ecdist.exchange_and_merge(eng, make_part, make_root, root=0, finalize_ranges=per_range)
            if m is not None:
                sizes.update(m.b.finalize())
        elif os.environ.get("ECB_ABLATE"):      # profiling-only builds of the kernel produce no ECs
            try:
                sizes.update(b.finalize())
            except ecb.EcbError:
                pass
        else:
            sizes.update(b.finalize())

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    if use_dist and world > 1:
        # set-up, not work: RCCL opens its point-to-point channels lazily, on the first message between two ranks
        ping = torch.zeros(world, dtype=torch.int64, device=red_dev)
        pong = torch.empty(world, dtype=torch.int64, device=red_dev)
        ops = []
        for r in range(world):
            if r != rank:
                ops += [dist.P2POp(dist.isend, ping[r:r + 1], r), dist.P2POp(dist.irecv, pong[r:r + 1], r)]
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        dist.all_reduce(ping)
        if not rehearsal:
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    b.profile(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    k_ms, k_launches, _ = b.profile_read()
    kernel_name = b.profile_kernel()              # (what libecb launched: k_stream.inc is compiled more than once, the library picks per batch)
    b.profile(False)
    # config 3 names bam2emase: the .h5 holds one CSC matrix per haplotype, so its device work is the step plus the transposition
    # of the finished CSR (ecb_csr_to_hapcsc_device).  Timed on its own, outside the K steps the headline is measured on.
    emase_ms = None
    if not use_dist and not multisample and not os.environ.get("ECB_ABLATE") and sizes.get("n_ecs"):
        ipt = torch.empty(sizes["n_ecs"] + 1, dtype=torch.int32, device=device)
        ixt = torch.empty(sizes["nnz_a"], dtype=torch.int32, device=device)
        dat = torch.empty(sizes["nnz_a"], dtype=torch.int32, device=device)

        def emase_step():
            step()
            b.export_device(ipt, ixt, dat)
            return ecb.csr_to_hapcsc(ipt, ixt, dat, T, H)

        emase_step()
        fence()
        per = []
        for _ in range(max(args.steps, 3)):          # (the median of single steps: this figure is the builder's own, and one allocator hiccup in five steps moved its mean by 2 ms)
            t0 = time.perf_counter()
            cptr, cidx = emase_step()
            fence()
            per.append((time.perf_counter() - t0) * 1e3)
        emase_ms = sorted(per)[len(per) // 2]
        emase_bits = int(cidx.numel())
        del ipt, ixt, dat, cptr, cidx
    exact = None
    if not use_dist and not multisample and not os.environ.get("ECB_ABLATE") and not os.environ.get("ECB_NO_VERIFY"):
        # outside the timed region: re-derive every read's target set and compare it with its EC's stored key
        b.reset()
        push()
        bad, skipped = b.verify_device_tiled(tiles, n_rec) if tiled else b.verify_device(rid, loc, hf)
        exact = {"reads_differing_from_their_ec_key": bad, "reads_on_the_long_read_path": skipped}
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt[0])
        tot = torch.tensor([st["records"], st["reads"], st["valid"]], dtype=torch.int64, device=red_dev)
        dist.all_reduce(tot)
        total_records, total_reads, total_valid = [int(x) for x in tot.tolist()]
    else:
        total_records, total_reads, total_valid = st["records"], st["reads"], st["valid"]

    if rank == 0:
        ms_per_step = dt * 1e3 / args.steps
        # dominant kernel: k_stream.  Algorithmic bytes per launch: every record tuple read
        # once (12 B) + one EC-slot id written per read (4 B).  This rank's launch, this rank's bytes.
        launches_per_step = max(k_launches / max(args.steps, 1), 1e-9)
        k_ms_per_launch = k_ms / max(k_launches, 1)
        alg_bytes = 12.0 * st["records"] + 4.0 * st["reads"]
        achieved = alg_bytes / (k_ms / max(args.steps, 1) * 1e-3) / 1e9      # GB/s over the kernel's own time
        # the whole step against SURVEY 8d's B_alg = 12 A + 4 R + 4 (E + 1) + 8 nnz(A): tuples -> CSR, everything in between
        E, nnz = sizes.get("n_ecs") or 0, sizes.get("nnz_a") or 0
        step_bytes = 12.0 * total_records + 4.0 * total_reads + 4.0 * (E + 1) + 8.0 * nnz
        if multisample:      # + the cell stream and N
            step_bytes += 4.0 * total_reads + 8.0 * ms_sizes.get("nnz_n", 0)
        step_achieved = step_bytes / (ms_per_step * 1e-3) / 1e9
        peak_copy, peak_read, peak_how = copy_peak(device) if not rehearsal else (None, None, None)
        sensors = None
        if world == 1 and not rehearsal and not os.environ.get("ECB_ABLATE"):
            try:
                sensors = sensors_under_load(step, fence, local)
            except Exception as e:      # (never the reason a bench line is missing)
                sensors = {"error": repr(e)}
        # HBM traffic: PMC counters need rocprofv3 around the process, so the figure comes from the committed pass of tools/tools_final.sh
        # (profiles/traffic_<workload>_n<N>.json) -- taken only when that pass profiled the kernel that ran here, and said with the build
        # it was measured on
        traffic, traffic_src, traffic_build = None, None, None
        tfile = os.path.join(ROOT, "profiles", "traffic_%s_n%d.json" % (args.workload, world))
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                traffic_build = "%s, commit %s" % (tj.get("round", "?"), tj.get("commit", "not recorded"))
                if tj.get("kernel") != kernel_name:
                    traffic_src = "refused: profiles/%s was measured on %r, this run launched %r" % (os.path.basename(tfile), tj.get("kernel"), kernel_name)
                else:
                    traffic, traffic_src = tj.get("hbm_bytes_per_launch"), "from_profiles: " + tj.get("source", "profiles/ (rocprofv3 --pmc, separate passes)")
            except Exception:
                traffic = None
        out = {
            "metric": "alignments/sec BAM->EC (100M PE reads, 8-hap) at 1/2/4/8 MI355X",
            "value": total_records / (ms_per_step * 1e-3),
            "unit": "alignments/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "step_emase_ms": emase_ms,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "layout": ("whole tiles: one buffer, 512 read ids | 512 loci | 512 haplotype/flag words per tile (ecb_push_device_tiled)" if tiled else
                       "three arrays in one allocation (ecb_push_device)"),
            "config": {"workload": desc, "reads": R, "loci": T, "haplotypes": H, "paired_end": paired,
                       "records": total_records, "valid_alignments": total_valid, "reads_with_alignments": total_reads,
                       "ecs": sizes.get("n_ecs"), "nnz_a": sizes.get("nnz_a"),
                       "multisample": None if not multisample else dict(cells=n_cells, files=64, minimum_count=min_count, triples=sizes.get("nnz_n"),
                                                                        cells_kept=ms_sizes.get("n_cells_kept"), ecs_kept=ms_sizes.get("n_ecs_kept"), nnz_n=ms_sizes.get("nnz_n")),
                       "sharding": "contiguous reads over %d GPU(s)%s%s" % (world, ", per-rank EC tables cut into key ranges, exchanged point-to-point over RCCL, merged and finalized per range, rows assembled on rank 0" if world > 1 else "",
                                                                              "; the merged ECs broadcast, every rank's reads reduced to (EC, cell, file) triples against them, rank 0 filters" if (world > 1 and multisample) else ""),
                       "generate_s": round(t_gen, 2), "exactness_pass": exact,
                       "step_emase": None if emase_ms is None else "step + CSR -> per-haplotype CSC (%d row indices) on the device, what bam2emase's .h5 holds: %.2f ms" % (emase_bits, emase_ms)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "traffic_measured_on": traffic_build,
                         "kernel": kernel_name,      # (ecb_profile_kernel: the compilation of csrc/k_stream.inc the library launched)
                         "kernel_ms_per_launch": k_ms_per_launch,
                         "launches_per_step": launches_per_step,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "step_achieved": step_achieved, "step_frac": step_achieved / HBM_PEAK_GBS,
                         "step_algorithmic_bytes": step_bytes,
                         "step_ms_besides_kernel": ms_per_step - k_ms_per_launch * launches_per_step,      # (partition-and-count, ranking, CSR emit, reset, host waits)
                         "peak_measured_copy": peak_copy, "peak_measured_read": peak_read, "peak_measured_how": peak_how,
                         "frac_of_measured_copy": (achieved / peak_copy) if peak_copy else None},
        }
        out["config"]["sensors_under_load"] = sensors
        if world == 1 and not multisample and not os.environ.get("ECB_ABLATE") and not args.no_cpu_baseline and args.workload in ("c3", "c2"):
            def step_from(*t_):
                b.reset()
                if tiled:
                    b.push_device_tiled(t_[0], n_rec)
                else:
                    b.push_device(*t_)
                b.finalize()
            try:
                spread = placement_spread(b, (tiles,) if tiled else (rid, loc, hf), step_from)
                out["roofline"]["kernel_ms_by_placement"] = spread
                out["roofline"]["kernel_ms_by_placement_note"] = ("k_stream ms per launch with the same tuples in this run's own allocation (first) and in %d further allocations made "
                                                                  "side by side (layout: %s).  As three arrays, where the tuples sit in HBM moves the kernel by up to 14 %% on one box, "
                                                                  "reproducibly per allocation (profiles/r04_stream_regions*.txt); as whole tiles by somewhat less "
                                                                  "(profiles/r04_tiles_against_arrays.txt).  The headline is the first one's, whatever it is" % (len(spread) - 1, args.layout))
            except Exception as e:
                out["roofline"]["kernel_ms_by_placement"] = {"error": repr(e)}
        if not use_dist and not args.no_cpu_baseline and not multisample:
            if tiled:                          # (the host-side legs -- the oracle, the Python baseline, the PCIe-inclusive push -- take arrays: copies)
                rid, loc, hf = tiles_to_arrays(tiles, n_rec)
            out["cpu_baseline"], oracle_result = cpu_baseline(rid, loc, hf, H, args.cpu_sample_reads, args.cpu_threads)
            if oracle_result is not None:      # the whole workload went through the oracle: hold the GPU's result to it, bit for bit
                ok, diff = parity_vs_oracle(b, push, oracle_result)
                out["parity_vs_oracle"] = ok
                out["config"]["parity_vs_oracle"] = "CSR A, counts and counters of the whole workload equal oracle/ec_oracle.c's" if ok else "DIFFERS: " + ", ".join(diff)
                parity_failed = not ok
                del oracle_result
            if args.workload == "c3" and not args.no_secondary:
                # the other BASELINE config of one GPU, the paralog stream and the short-read stream, on this box in this run
                out["secondary"] = {}
                for name in ("c2", "c3r", "dip"):
                    try:
                        out["secondary"][name] = secondary(name, device, local, args.secondary_steps, True, args.layout)
                        if out["secondary"][name].get("parity_vs_oracle") is False:
                            parity_failed = True
                    except Exception as e:      # (a secondary line never costs the headline)
                        out["secondary"][name] = {"error": repr(e)}
            out["cpu_baseline_py"] = cpu_baseline_py(rid, loc, hf, H, min(args.py_sample_reads, st["reads"]))
            out["h2d_inclusive"] = h2d_inclusive(b, rid, loc, hf, 400_000_000)
            import tempfile
            with tempfile.TemporaryDirectory() as td:
                c1, c2 = WORKLOADS["c1"], WORKLOADS["c2"]
                out["e2e_from_bam"] = {"config1": e2e_from_bam((c1[0], c1[1], c1[2], c1[3]), c1[0], td),
                                       "config2_slice": e2e_from_bam((c2[0], c2[1], c2[2], c2[3]), args.e2e_slice_reads, td)}
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if parity_failed:
        raise SystemExit("bench.py: the GPU's result differs from the oracle's on the whole workload (see config.parity_vs_oracle)")


if __name__ == "__main__":
    main()
