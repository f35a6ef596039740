"""Config-4-shaped run for profiling: R paired-end reads, 5 000 cell barcodes over 64 files, the whole multisample path on one GPU
(stream kernel, triple reduce with the repo's radix sort, ecb_ms_filter).  usage: python tools/run_c4.py [reads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from alntools_amd import ecb, synth
R = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
T, H = 80_000, 8
dev = torch.device("cuda:0")
rid, loc, hf, st = bench.generate_shard(synth.SynthSpec(R, T, H, paired=True), 0, R, dev)
n = st["reads"]
g = torch.arange(n, dtype=torch.int64, device=dev)
x = ((g * 0x9E3779B97F4A7C15) >> 20) & 0x7FFFFFFFFFF
cell = x % 5000
_, inv = torch.unique(cell, return_inverse=True)                      # (ids need not be in first-seen order for a profile)
meta = (inv | (((g * 64) // n) << 22)).to(torch.int32).cpu().numpy().view(np.uint32)
with ecb.EcBuilder(T, H, multisample=True, ec_capacity=1 << 24) as b:
    for rep in range(2):
        b.reset()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        b.push_device(rid, loc, hf); b.push_cells(meta, 0)
        s = b.finalize(); t1 = time.perf_counter()
        f = b.ms_filter(5000, 1000); t2 = time.perf_counter()
    print("c4-shaped: %d reads, %d records, %d ECs, %d triples; push+finalize (EC build + triple reduce) %.1f ms, ms_filter %.1f ms; kept %d cells, %d ECs, nnz N %d"
          % (n, st["records"], s["n_ecs"], s["nnz_n"], (t1 - t0) * 1e3, (t2 - t1) * 1e3, len(f["kept_cells"]), len(f["indptrA"]) - 1, len(f["dataN"])))
