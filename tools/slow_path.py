"""GPU box: what the long-read path costs.  A single-end stream of 20 M reads of ~16 records in which a given share of the reads
are long: `n_long_records` records on distinct loci, i.e. more (locus, mask) entries than a wave carries from tile to tile (CMAX),
so k_stream hands them to k_slow (one workgroup per read).  Prints step time and k_slow's share next to the all-short stream.
usage: python tools/slow_path.py [reads] [records per long read]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alntools_amd import ecb
R = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 700
T, H = 40_000, 8
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(20260101)


def stream(share):
    """reads of 16 records on 2 loci x 8 haplotypes; every round(1/share)-th read is long (NL records on NL distinct loci)"""
    n = torch.full((R,), 16, dtype=torch.int64, device=dev)
    long_ = torch.zeros(R, dtype=torch.bool, device=dev)
    if share > 0:
        long_[torch.arange(0, R, round(1 / share), device=dev)] = True
        n[long_] = NL
    rid = torch.repeat_interleave(torch.arange(R, dtype=torch.int32, device=dev), n)
    start = torch.cumsum(n, 0) - n
    k = torch.arange(rid.numel(), dtype=torch.int64, device=dev) - start[rid.long()]          # record index within its read
    base = torch.randint(0, T - NL - 2, (R,), generator=g, device=dev)[rid.long()]
    is_long = long_[rid.long()]
    loc = torch.where(is_long, base + k, base + (k >> 3)).to(torch.int32)
    hap = torch.where(is_long, k & 7, k & 7)
    hf = (hap << 16).to(torch.int32)
    pad = (-rid.numel()) % 4                                                                 # (nothing to pad: device streams only need aligned bases)
    return rid, loc, hf, int(long_.sum())


with ecb.EcBuilder(T, H, device=0, ec_capacity=1 << 25) as b:
    for share in (0.0, 0.001, 0.01):
        rid, loc, hf, nlong = stream(share)
        for rep in range(2):
            b.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
            b.push_device(rid, loc, hf); torch.cuda.synchronize(); t1 = time.perf_counter()
            s = b.finalize(); t2 = time.perf_counter()
        bad, n_slow = b.verify_device(rid, loc, hf)
        print("long reads %.1f %% (%d of %d, %d records each): %d records, %d ECs; push %.2f ms + finalize %.2f ms = %.2f G records/s; "
              "exactness pass: %d differing, %d reads took the long-read path" % (
                  share * 100, nlong, R, NL, rid.numel(), s["n_ecs"], (t1 - t0) * 1e3, (t2 - t1) * 1e3, rid.numel() / (t2 - t0) / 1e9, bad, n_slow), flush=True)
        del rid, loc, hf
