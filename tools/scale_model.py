#!/usr/bin/env python
"""GPU box (one GPU): time the pieces of the N-GPU step on ONE card to see where a strong-scaling run spends its time.
For N in 1, 2, 4, 8 the c3 stream is cut into N contiguous read shards; every shard's rank-side work (push, counts,
export by key range), the per-range merges (one rank's share), then either the range's own finalize and the root's
assembly of the finished ranges (what dist.py does for single-sample runs) or the range's export and the root's adopt +
finalize (multisample runs) are timed separately, next to the older whole-table-to-root protocol.  Transfers are not modelled.
usage: python tools/scale_model.py [workload]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from alntools_amd import ecb, synth, dist as ecdist

w = sys.argv[1] if len(sys.argv) > 1 else "c3"
NS = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (1, 2, 4, 8)
SIMPLE = len(sys.argv) <= 3 or sys.argv[3] != "ranges"
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


asm = ecb.EcBuilder(T, H, device=0, ec_capacity=1 << 20, arena_capacity=1 << 20)
aeng = ecdist.GpuEngine(asm, dev)
for N in NS:
    # (the handle that merges one key range, sized as bench.py sizes it: room for every arriving entry, about one shard's ECs)
    part = ecb.EcBuilder(T, H, device=0, ec_capacity=1 << (23 if N >= 4 else 24), arena_capacity=1 << 26)
    peng = ecdist.GpuEngine(part, dev)
    for rep in range(2):
        rank_ms, pieces, base, tot = [], [], 0, [0, 0, 0]
        for r in range(N):
            lo = int(torch.searchsorted(rid, torch.tensor([r * st["reads"] // N], dtype=torch.int32, device=dev))[0]) if r else 0
            hi = int(torch.searchsorted(rid, torch.tensor([(r + 1) * st["reads"] // N], dtype=torch.int32, device=dev))[0]) if r < N - 1 else rid.numel()
            s = [t[lo:hi].clone() for t in (rid, loc, hf)]
            s[0] -= r * st["reads"] // N
            t0 = sync()
            b.reset()
            b.push_device(*s)
            t1 = sync()
            ne, npairs, nreads = b.table_sizes()
            t2 = sync()
            pieces.append(eng.table_export_parts(base, N))
            t3 = sync()
            a, v, _ = b.counters()
            rank_ms.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
            base += nreads
            tot = [tot[0] + a, tot[1] + v, tot[2] + nreads]
            del s
        root.reset()
        part_ms, fin_ms, adopt_ms, moved, moved_fin, merged, done = [], [], 0.0, [], [], [], []
        for q in range(N):
            mine = [(ent[eoff[q] * 4:eoff[q + 1] * 4], eoff[q + 1] - eoff[q], prs[poff[q]:poff[q + 1]], poff[q + 1] - poff[q])
                    for ent, prs, eoff, poff in pieces if eoff[q + 1] > eoff[q]]
            t0 = sync()
            part.reset()
            peng.table_merge_many(mine)
            tm = sync()
            pe_n, pp_n, _ = part.table_sizes()
            pe, pp = peng.table_export(0)
            t1 = sync()
            merged.append((pe.clone(), pe_n, pp.clone(), pp_n))      # (what arrives at the root: all ranges, adopted in one call as dist.py does)
            part_ms.append(((tm - t0) * 1e3, (t1 - tm) * 1e3))
            moved.append((pe_n * 32 + pp_n * 8) / 1e6)
            part.reset()                                              # the other ending: the range is finalized where it was merged
            peng.table_merge_many(mine)
            t0 = sync()
            done.append(peng.finalize_range(*tot))
            fin_ms.append((sync() - t0) * 1e3)
            moved_fin.append(done[-1][0].numel() * 4 / 1e6)
        asm.reset()
        t0 = sync()
        sz_a = aeng.assemble_ranges(done, *tot)
        asm_ms = (sync() - t0) * 1e3
        del done
        # the multisample ending since round 4: the same per-range finalize and assembly; what the root adds for the second exchange is the
        # ECs' hashes, taken off the assembled rows (ecb_export_ec_keys_device on a handle without a table)
        keys = torch.empty(sz_a["n_ecs"], dtype=torch.int64, device=dev)
        t0 = sync()
        asm.export_ec_keys_device(keys)
        keys_ms = (sync() - t0) * 1e3
        del keys
        t1 = sync()
        reng.table_adopt_many(merged)
        adopt_ms = (sync() - t1) * 1e3
        del merged
        root.add_counters(*tot)
        t5 = sync()
        sz = root.finalize()
        t6 = sync()
        sent = [sum((p[2][q + 1] - p[2][q]) * 32 + (p[3][q + 1] - p[3][q]) * 8 for q in range(N) if q != r) / 1e6 for r, p in enumerate(pieces)]
        del pieces
    rk = max(sum(x) for x in rank_ms)
    mg = max(x[0] for x in part_ms)
    assert sz_a == sz
    print("ranges N=%d: rank-side max %.2f ms (push %.2f, counts %.2f, cut+export %.2f) | my range: merge %.2f ms, then finalize %.2f ms | root: assemble %.2f ms | "
          "sent per rank %.0f MB, to root %.0f MB (xGMI at 153 GB/s per link, every peer on its own link: %.2f + %.2f ms, not in the model) | "
          "model step %.2f ms  ECs %d" % (
              N, rk, max(x[0] for x in rank_ms), max(x[1] for x in rank_ms), max(x[2] for x in rank_ms), mg, max(fin_ms), asm_ms,
              max(sent), sum(moved_fin[1:]), (max(sent) / max(N - 1, 1)) / 153.0, (max(moved_fin[1:]) if N > 1 else 0.0) / 153.0,
              rk + mg + max(fin_ms) + asm_ms, sz["n_ecs"]), flush=True)
    print("   (multisample ending, round 4: finalize per range + assembly + the ECs' hashes off the assembled rows) N=%d: hashes %.2f ms | model step up to the second exchange %.2f ms" % (
        N, keys_ms, rk + mg + max(fin_ms) + asm_ms + keys_ms), flush=True)
    print("   (multisample ending until round 3, kept as finalize_ranges=False) N=%d: my range: merge %.2f + export %.2f ms | root: adopt %.2f + finalize %.2f ms | to root %.0f MB (%.2f ms) | model step %.2f ms" % (
        N, mg, max(x[1] for x in part_ms), adopt_ms, (t6 - t5) * 1e3, sum(moved[1:]), (max(moved[1:]) if N > 1 else 0.0) / 153.0,
        rk + max(sum(x) for x in part_ms) + adopt_ms + (t6 - t5) * 1e3), flush=True)
    part.close()

for N in (NS if SIMPLE else ()):
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
