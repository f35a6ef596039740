#!/usr/bin/env python
"""Where the tuple streams sit in HBM: several candidate arenas in one process (all allocated at once, so that they are different
memory), the same config-3 tuples copied into each; per arena the rate of a pure read sweep over it (tools/micro/copy_peak.hip) and
k_stream's time from it.  Does the sweep's rate predict the kernel's time?  (profiles/r04_stream_placement*.txt: the kernel's time on
one box moves by 8 - 10 % with where the streams are.)"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from alntools_amd import ecb  # noqa: E402
from r04_place import read_rate, timed  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
    n_arenas = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device("cuda:0")
    R, T, H, paired, _ = bench.WORKLOADS[wl]
    spec = bench.workload_spec(wl)
    rid, loc, hf, st = bench.generate_shard(spec, 0, R, dev)
    n = rid.numel()
    stride = (n * 4 + (2 << 20) - 1) // (2 << 20) * (2 << 20) // 4
    slack = (48 << 20) // 4
    arenas = [None] + [torch.empty(3 * stride + slack, dtype=torch.int32, device=dev) for _ in range(n_arenas - 1)]
    staggers = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
    rounds = [x for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else []       # ECB_ROUNDS[:ECB_MIN_TILES] settings: slices per wave of the launch
    cap = 1 << (25 if wl == "c3r" else 24 if wl.startswith("c3") else 22)
    if rounds:                                   # slice geometry against placement: a handle per setting (the launch shape is read once per handle)
        for i, a in enumerate(arenas):
            if a is None:
                views = (rid, loc, hf)
            else:
                views = (a[:n], a[stride:stride + n], a[2 * stride:2 * stride + n])
                for v, src in zip(views, (rid, loc, hf)):
                    v.copy_(src)
                torch.cuda.synchronize()
            for setting in rounds:
                r_, _, m_ = setting.partition(":")
                os.environ["ECB_ROUNDS"] = r_
                if m_:
                    os.environ["ECB_MIN_TILES"] = m_
                else:
                    os.environ.pop("ECB_MIN_TILES", None)
                with ecb.EcBuilder(T, H, device=0, ec_capacity=cap, arena_capacity=1 << 26) as b:
                    b.hint_reads(st["reads"])
                    k, s = timed(b, *views)
                print("arena %d at %x  slices per wave %s: k_stream %.3f ms  step %.2f ms" % (i, views[0].data_ptr(), setting, k, s), flush=True)
        return
    with ecb.EcBuilder(T, H, device=0, ec_capacity=cap, arena_capacity=1 << 26) as b:
        b.hint_reads(st["reads"])
        for rep in range(2):
            for i, a in enumerate(arenas):
                for stag in (staggers if a is not None else [0]):
                    if a is None:
                        views = (rid, loc, hf)
                    else:
                        w = stag // 4                  # stream s starts s * stagger bytes further on than it would
                        views = (a[:n], a[stride + w:stride + w + n], a[2 * stride + 2 * w:2 * stride + 2 * w + n])
                        if rep == 0 or len(staggers) > 1:
                            for v, src in zip(views, (rid, loc, hf)):
                                v.copy_(src)
                            torch.cuda.synchronize()
                    rr = read_rate(views) if len(staggers) == 1 else None
                    k, s = timed(b, *views)
                    print("rep %d arena %d at %x stagger %d: read sweep %s GB/s (mean %.0f)   k_stream %.3f ms  step %.2f ms" % (
                        rep, i, views[0].data_ptr(), stag, rr, sum(rr) / 3.0 if rr else 0.0, k, s), flush=True)


if __name__ == "__main__":
    main()
