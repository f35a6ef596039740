"""GPU box: is the cost of founding ECs a start-up transient (thousands of waves founding the same hot ECs at once)?
Push a short prefix of the stream first (untimed), then time k_stream on the rest."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from alntools_amd import ecb, synth
R, T, H, paired, _ = bench.WORKLOADS["c3"]
dev = torch.device("cuda", 0)
rid, loc, hf, st = bench.generate_shard(synth.SynthSpec(R, T, H, paired=paired), 0, R, dev)
b = ecb.EcBuilder(T, H, device=0, ec_capacity=1 << 24, arena_capacity=1 << 26)
for pre in (0, 100_000, 1_000_000, 4_000_000, 0):
    cut = int(torch.searchsorted(rid, torch.tensor([pre], dtype=torch.int32, device=dev))[0]) if pre else 0
    cut -= cut % 4
    while cut and int(rid[cut]) == int(rid[cut - 1]): cut += 4      # keep it simple: need a read boundary on a 16-byte boundary
    ok = cut == 0 or int(rid[cut]) != int(rid[cut - 1])
    if not ok: continue
    b.reset()
    if cut: b.push_device(rid[:cut], loc[:cut], hf[:cut])
    torch.cuda.synchronize()
    e0 = b.counters()
    b.profile(True)
    b.push_device(rid[cut:], loc[cut:], hf[cut:])
    torch.cuda.synchronize()
    ms, n, _ = b.profile_read()
    b.profile(False)
    print("prefix %d reads (%d records): k_stream on the rest %.3f ms (%d launches) = %.3f ms per G records" % (pre, cut, ms, n, ms / ((rid.numel() - cut) / 1e9)), flush=True)
