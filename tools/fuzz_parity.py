#!/usr/bin/env python
"""GPU box: many more random streams than the test suite runs (tests/test_gpu_parity.py::_random_stream), each through the C ABI --
host pushes in random batch sizes, one device-resident push, a tiny EC table that has to grow, the exactness pass, and the
per-range protocol on one card (2 - 5 shards) -- against the C oracle, bit for bit.  usage: python tools/fuzz_parity.py [first_seed] [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from alntools_amd import ecb, dist as ecdist
from oracle import c_oracle
from tests.test_gpu_parity import _random_stream, _check, _run_host

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)
t0 = time.time()
for seed in range(first, first + n_cases):
    rng = np.random.RandomState(seed)
    n_haps = int(rng.choice([1, 2, 3, 8, 16, 31]))
    n_loci = int(rng.choice([3, 17, 300, 5000, 1 << 16, 1 << 22, (1 << 26) - 3]))
    max_len = int(rng.choice([2, 5, 12, 40, 90, 300, 700]))
    p_inv = float(rng.choice([0.0, 0.05, 0.3, 0.6]))
    mode = str(rng.choice(["wide", "strided", "near"]))
    n_reads = int(rng.choice([1, 7, 500, 3000, 6000])) if max_len <= 90 else int(rng.choice([50, 400]))
    t = _random_stream(seed, n_reads, n_loci, n_haps, max_len, p_inv, mode)
    try:
        exp = c_oracle.ec_from_tuples(t["read_id"], t["locus"], t["hapflag"], n_haps, threads=2)
    except ValueError:                             # (no valid record at all: the reference fails there, and so does ecb_finalize)
        print("seed %d: no valid record, skipped" % seed, flush=True)
        continue
    tag = "seed %d: reads %d loci %d haps %d max_len %d p_inv %.2f %s (%d records, %d ECs)" % (
        seed, n_reads, n_loci, n_haps, max_len, p_inv, mode, len(t["read_id"]), len(exp["count"]))
    # host pushes: whole, and in random batches; a table that starts too small
    for batch, kw in ((None, {}), (int(rng.randint(1, 2000)), {}), (int(rng.randint(100, 5000)), dict(ec_capacity=64))):
        out, sizes = _run_host(t, n_loci, n_haps, batch=batch, **kw)
        _check(out, sizes, exp)
    # device-resident push + the exactness pass
    d = [torch.from_numpy(t[k].view(np.int32)).to(dev) for k in ("read_id", "locus", "hapflag")]
    for k in ("ECB_FORCE_PAR", "ECB_NO_PAR"):          # the compilation of the stream kernel: the library's choice, ks_par forced, ks_par ruled out -- in turn
        os.environ.pop(k, None)
    if seed % 3:
        os.environ["ECB_FORCE_PAR" if seed % 3 == 1 else "ECB_NO_PAR"] = "1"
    with ecb.EcBuilder(n_loci, n_haps) as b:
        if seed & 1:                                   # every other case with the stream's reads bounded up front (one wait per push)
            b.hint_reads(int(exp["n_reads"]) if "n_reads" in exp else len(t["read_id"]))
        b.push_device(*d)
        sizes = b.finalize(); out = b.export()
        _check(out, sizes, exp)
        b.reset(); b.push_device(*d)                   # (a handle that keeps its per-read slot ids across the reset)
        bad, _ = b.verify_device(*d)
        assert bad == 0, tag
    # the key-range protocol on one card: shards cut at read boundaries, merged per range, finalized per range, assembled
    rid = t["read_id"].astype(np.int64); rid[rid == 0xFFFFFFFF] = -1
    R = int(rid.max()) + 1
    world = int(rng.randint(2, 6))
    if R >= world:
        cuts, bases, tot = [], [], [0, 0, 0]
        for r in range(world):
            lo, hi = r * R // world, (r + 1) * R // world
            m = (rid >= lo) & (rid < hi) if r else (rid < hi)
            sb = ecb.EcBuilder(n_loci, n_haps, ec_capacity=256)
            local = np.where(rid[m] < 0, 0xFFFFFFFF, rid[m] - lo).astype(np.uint32)
            sb.push(local, t["locus"][m], t["hapflag"][m])
            eng = ecdist.GpuEngine(sb, dev)
            cuts.append(eng.table_export_parts(0, world)); bases.append(tot[2])
            a, v, n = eng.counters(); tot = [tot[0] + a, tot[1] + v, tot[2] + n]
            sb.close()
        pieces = []
        for q in range(world):
            part = ecdist.GpuEngine(ecb.EcBuilder(n_loci, n_haps, ec_capacity=256), dev)
            part.table_merge_many([(part.table_rebase(ent[eo[q] * 4:eo[q + 1] * 4], eo[q + 1] - eo[q], base), eo[q + 1] - eo[q], prs[po[q]:], po[q + 1] - po[q])
                                   for (ent, prs, eo, po), base in zip(cuts, bases) if eo[q + 1] > eo[q]])
            pieces.append(part.finalize_range(*tot)); part.b.close()
        root = ecdist.GpuEngine(ecb.EcBuilder(n_loci, n_haps, ec_capacity=64), dev)
        s = root.assemble_ranges([p for p in pieces if p[1]], *tot)
        _check(root.b.export(), s, exp)
        root.b.close()
    print(tag, "ok", flush=True)
print("%d cases in %.0f s" % (n_cases, time.time() - t0))
