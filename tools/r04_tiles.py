#!/usr/bin/env python
"""Three arrays against one buffer of whole tiles (ecb_push_device_tiled), several allocations of each side by side in one process: k_stream's
time per allocation.  Does the tile layout take the dependence on where the tuples sit out of the kernel?  usage: r04_tiles.py [workload] [n]"""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from alntools_amd import ecb  # noqa: E402


def timed(b, push, steps=3):
    b.reset(); push(); s = b.finalize()
    b.profile(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        b.reset(); push(); b.finalize()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ms, n, _ = b.profile_read()
    b.profile(False)
    return ms / max(n, 1), dt * 1e3, s


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dev = torch.device("cuda:0")
    R, T, H, paired, _ = bench.WORKLOADS[wl]
    spec = bench.workload_spec(wl)
    rid, loc, hf, st = bench.generate_shard(spec, 0, R, dev)
    n = rid.numel()
    stride = (n * 4 + (2 << 20) - 1) // (2 << 20) * (2 << 20) // 4
    nt = (n + 511) // 512
    soa = [None] + [torch.empty(3 * stride, dtype=torch.int32, device=dev) for _ in range(k - 1)]
    til = [torch.zeros(nt * 1536, dtype=torch.int32, device=dev) for _ in range(k)]
    cap = 1 << (25 if wl == "c3r" else 24 if wl.startswith("c3") else 22)
    with ecb.EcBuilder(T, H, device=0, ec_capacity=cap, arena_capacity=1 << 26) as b:
        b.hint_reads(st["reads"])
        ref = None
        for rep in range(2):
            for i in range(k):
                a = soa[i]
                if a is None:
                    v = (rid, loc, hf)
                else:
                    v = (a[:n], a[stride:stride + n], a[2 * stride:2 * stride + n])
                    if rep == 0:
                        for d, src in zip(v, (rid, loc, hf)):
                            d.copy_(src)
                ks, ss, s = timed(b, lambda: b.push_device(*v))
                ref = ref or s
                assert s == ref
                if rep == 0:
                    ecb.tile_tuples(rid, loc, hf, out=til[i])
                    torch.cuda.synchronize()
                kt, stt, s2 = timed(b, lambda: b.push_device_tiled(til[i], n))
                assert s2 == ref, (s2, ref)
                print("rep %d allocation %d: three arrays at %x: k_stream %.3f ms (step %.2f)   whole tiles at %x: k_stream %.3f ms (step %.2f)   %s" % (
                    rep, i, v[0].data_ptr(), ks, ss, til[i].data_ptr(), kt, stt, b.profile_kernel()), flush=True)
        b.reset(); b.push_device_tiled(til[0], n)
        print("exactness pass over the tiled buffer:", b.verify_device_tiled(til[0], n), " sizes:", ref)


if __name__ == "__main__":
    main()
