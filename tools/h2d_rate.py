#!/usr/bin/env python
"""GPU box: PCIe-inclusive rate of the host-pointer path (ecb_push from numpy arrays), for DESIGN.md section 6."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from alntools_amd import ecb, synth

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
PE = len(sys.argv) > 2 and sys.argv[2] == "pe"
spec = synth.SynthSpec(R, 80_000 if PE else 40_000, 8, paired=PE)
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
rid, loc, hf, st = bench.generate_shard(spec, 0, spec.n_reads, torch.device("cuda:0"))
h = {"read_id": rid.cpu().numpy().view(np.uint32), "locus": loc.cpu().numpy().view(np.uint32), "hapflag": hf.cpu().numpy().view(np.uint32)}
del rid, loc, hf
torch.cuda.empty_cache()
n = len(h["read_id"])
with ecb.EcBuilder(spec.n_loci, spec.n_haps, max_batch_records=1 << 24, ec_capacity=1 << 24) as b:
    for it in range(3):
        b.reset()
        t0 = time.perf_counter()
        b.push(h["read_id"], h["locus"], h["hapflag"])
        s = b.finalize()
        dt = time.perf_counter() - t0
        print("host-pointer path: %d records in %.1f ms = %.2f G records/s (%.1f GB/s of tuples), %d ECs" %
              (n, dt * 1e3, n / dt / 1e9, 12 * n / dt / 1e9, s["n_ecs"]))
