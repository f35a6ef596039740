"""Config-5-shaped run for profiling: config 3's stream with positions -> EC build + ranges, then CSR(bitmask) -> per-haplotype CSC
-> CSR on the 12 M-nnz result (the device half of ec2emase / emase2ec).  usage: python tools/run_c5.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from alntools_amd import ecb, synth
R, T, H, paired, _ = bench.WORKLOADS["c3"]
dev = torch.device("cuda:0")
rid, loc, hf, st = bench.generate_shard(synth.SynthSpec(R, T, H, paired=paired), 0, R, dev)
pos = ((torch.arange(rid.numel(), dtype=torch.int64, device=dev) * 1103515245 + 12345) % 4999).to(torch.int32)
with ecb.EcBuilder(T, H, ec_capacity=1 << 24, track_ranges=True) as b:
    for rep in range(2):
        b.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
        b.push_device(rid, loc, hf, pos); s = b.finalize(); t1 = time.perf_counter()
    out = b.export(); rng = b.export_ranges()
ip, ix, da = (torch.from_numpy(out[k]).to(dev) for k in ("indptrA", "indicesA", "dataA"))
for rep in range(2):
    torch.cuda.synchronize(); t2 = time.perf_counter()
    cptr, cidx = ecb.csr_to_hapcsc(ip, ix, da, T, H); torch.cuda.synchronize(); t3 = time.perf_counter()
    ip2, ix2, da2 = ecb.hapcsc_to_csr(cptr, cidx, s["n_ecs"]); torch.cuda.synchronize(); t4 = time.perf_counter()
assert torch.equal(ip2, ip) and torch.equal(ix2, ix) and torch.equal(da2, da)
print("c5-shaped: %d records with positions: EC build + ranges %.1f ms; %d ECs, nnz %d, %d set bits: csr->hapcsc %.2f ms, hapcsc->csr %.2f ms; round trip exact"
      % (st["records"], (t1 - t0) * 1e3, s["n_ecs"], s["nnz_a"], cidx.numel(), (t3 - t2) * 1e3, (t4 - t3) * 1e3))
