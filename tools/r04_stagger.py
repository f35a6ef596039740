#!/usr/bin/env python
"""Does k_stream's time on config 3 depend on how the three tuple streams sit RELATIVE to each other?  One allocation holds all three;
stream s starts at s * (size rounded up to 2 MiB) + s * stagger bytes, for a list of staggers, at two shifts of the whole arrangement.
(tools/r04_place.py showed that moving the streams changes the time by 8 % with everything else fixed.)"""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from alntools_amd import ecb  # noqa: E402
from r04_place import timed  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
    dev = torch.device("cuda:0")
    R, T, H, paired, _ = bench.WORKLOADS[wl]
    spec = bench.workload_spec(wl)
    rid, loc, hf, st = bench.generate_shard(spec, 0, R, dev)
    n = rid.numel()
    words = ((n * 4 + (2 << 20) - 1) // (2 << 20)) * (2 << 20) // 4          # a stream's stride: its size rounded up to 2 MiB
    slack = (64 << 20) // 4
    big = torch.empty(3 * words + 4 * slack, dtype=torch.int32, device=dev)
    print("%d records per stream; arena at %x" % (n, big.data_ptr()), flush=True)
    with ecb.EcBuilder(T, H, device=0, ec_capacity=1 << 24, arena_capacity=1 << 26) as b:
        b.hint_reads(st["reads"])
        k, s = timed(b, rid, loc, hf)
        print("as generated (three allocations at %x %x %x): k_stream %.3f ms  step %.2f ms" % (rid.data_ptr(), loc.data_ptr(), hf.data_ptr(), k, s), flush=True)
        for shift in (0, 1 << 20, (8 << 20) + 4096):                   # bytes: the whole arrangement moved
            for stag in (0, 256, 4096, 65536, 1 << 20, (2 << 20) + 4096, (16 << 20) + 65536):
                views = []
                for i, src in enumerate((rid, loc, hf)):
                    off = (shift + i * stag) // 4 + i * words
                    v = big[off:off + n]
                    v.copy_(src)
                    views.append(v)
                torch.cuda.synchronize()
                k, s = timed(b, *views)
                print("shift %9d  stagger %9d  k_stream %.3f ms  step %.2f ms" % (shift, stag, k, s), flush=True)


if __name__ == "__main__":
    main()
