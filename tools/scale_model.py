#!/usr/bin/env python
"""GPU box (one GPU): time the pieces of the N-GPU step on ONE card to see where a strong-scaling run spends its time.
For N in 1, 2, 4, 8 the c3 stream is cut into N contiguous read shards; every shard's rank-side work (push, counts,
table export) and the root-side work (N merges, finalize) are timed separately.  Transfers are not modelled.
usage: python tools/scale_model.py [workload]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from alntools_amd import ecb, synth, dist as ecdist

w = sys.argv[1] if len(sys.argv) > 1 else "c3"
R, T, H, paired, _ = bench.WORKLOADS[w]
dev = torch.device("cuda", 0)
spec = synth.SynthSpec(R, T, H, paired=paired)
rid, loc, hf, st = bench.generate_shard(spec, 0, R, dev)
cap = 1 << 24
b = ecb.EcBuilder(T, H, device=0, ec_capacity=cap, arena_capacity=1 << 26)
root = ecb.EcBuilder(T, H, device=0, ec_capacity=cap, arena_capacity=1 << 26)
eng, reng = ecdist.GpuEngine(b, dev), ecdist.GpuEngine(root, dev)


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


for N in (1, 2, 4, 8):
    for rep in range(2):
        root.reset()
        rank_ms, merge_ms, tabs, base = [], [], [], 0
        tot = [0, 0, 0]
        for r in range(N):
            lo = int(torch.searchsorted(rid, torch.tensor([r * st["reads"] // N], dtype=torch.int32, device=dev))[0]) if r else 0
            hi = int(torch.searchsorted(rid, torch.tensor([(r + 1) * st["reads"] // N], dtype=torch.int32, device=dev))[0]) if r < N - 1 else rid.numel()
            s = [t[lo:hi].clone() for t in (rid, loc, hf)]
            s[0] -= r * st["reads"] // N            # a fresh handle counts its reads from 0
            t0 = sync()
            b.reset()
            b.push_device(*s)
            t1 = sync()
            ne, npairs, nreads = b.table_sizes()
            t2 = sync()
            ent, prs = eng.table_export(base)
            t3 = sync()
            a, v, _ = b.counters()
            rank_ms.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
            reng.table_merge(ent, ne, prs, npairs)
            t4 = sync()
            merge_ms.append((t4 - t3) * 1e3)
            base += nreads
            tot = [tot[0] + a, tot[1] + v, tot[2] + nreads]
            tabs.append((ne, npairs))
            del s, ent, prs
        root.add_counters(*tot)
        t5 = sync()
        sz = root.finalize()
        t6 = sync()
    rk = max(sum(x) for x in rank_ms)
    print("N=%d rank-side max %.2f ms (push %.2f, counts %.2f, export %.2f) | root merges %s = %.2f ms, finalize %.2f ms | "
          "tables %s | model step %.2f ms  ECs %d" % (
              N, rk, max(x[0] for x in rank_ms), max(x[1] for x in rank_ms), max(x[2] for x in rank_ms),
              ["%.2f" % m for m in merge_ms], sum(merge_ms), (t6 - t5) * 1e3,
              ["%.1fMB" % ((e * 32 + p * 8) / 1e6) for e, p in tabs], rk + sum(merge_ms) + (t6 - t5) * 1e3, sz["n_ecs"]), flush=True)
