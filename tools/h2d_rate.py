#!/usr/bin/env python
"""GPU box: PCIe-inclusive rate of the host-pointer path (ecb_push from numpy arrays), for DESIGN.md section 6."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from alntools_amd import ecb, synth

spec = synth.SynthSpec(4_000_000, 40_000, 8)
import torch
g = synth.generate(spec, 0, spec.n_reads, device=torch.device("cuda:0"))
h = {k: g[k].cpu().numpy().view(np.uint32) for k in ("read_id", "locus", "hapflag")}
n = len(h["read_id"])
with ecb.EcBuilder(spec.n_loci, spec.n_haps, max_batch_records=1 << 24) as b:
    for it in range(3):
        b.reset()
        t0 = time.perf_counter()
        b.push(h["read_id"], h["locus"], h["hapflag"])
        s = b.finalize()
        dt = time.perf_counter() - t0
        print("host-pointer path: %d records in %.1f ms = %.2f G records/s (%.1f GB/s of tuples), %d ECs" %
              (n, dt * 1e3, n / dt / 1e9, 12 * n / dt / 1e9, s["n_ecs"]))
