#!/usr/bin/env python
"""tools/micro/stream3.hip over several allocations side by side: the read side of k_stream with its shape as parameters (unit per step,
streams read together, slice length, waves per SIMD), GB/s per allocation.  Which part of the shape makes the rate depend on where the
streams sit?  usage: python tools/r04_stream3.py [n_arenas] [GiB per stream]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def main():
    n_arenas = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    gib = float(sys.argv[2]) if len(sys.argv) > 2 else 12.27          # config 3: 13.18 GB per stream
    lib = ctypes.CDLL(os.path.join(ROOT, "tools", "micro", "bin", "libstream3.so"))
    lib.stream3_run.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint64, ctypes.c_int,
                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    dev = torch.device("cuda:0")
    nbytes = int(gib * (1 << 30)) // (2 << 20) * (2 << 20)
    arenas = [torch.empty(3 * nbytes // 4, dtype=torch.int32, device=dev) for _ in range(n_arenas)]
    for a in arenas:
        a.fill_(1)
    ctr = torch.zeros(2, dtype=torch.int64, device=dev)
    out = torch.zeros(1 << 16, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    shapes = [  # (streams, 1 KB units per step, slice bytes, waves per SIMD)
        (3, 2, 104 << 10, 5),      # k_stream on config 3: three streams, 2 KB per step, ~52 tiles per slice, five waves per SIMD
        (1, 2, 104 << 10, 5), (3, 1, 104 << 10, 5), (3, 4, 104 << 10, 5), (3, 8, 104 << 10, 5), (3, 8, 104 << 10, 2),
        (3, 2, 16 << 10, 5), (3, 2, 1 << 20, 5), (3, 2, 8 << 20, 5), (3, 2, 104 << 10, 8), (3, 4, 1 << 20, 8), (1, 8, 4 << 20, 8),
        # one stream read in steps of 6 KB: what a layout of whole tiles -- a tile's 2 KB of read ids, loci and haplotype/flag words side by side --
        # would make of k_stream's reads (slices of the same 52 tiles)
        (1, 6, 312 << 10, 5), (1, 6, 1248 << 10, 5), (1, 3, 312 << 10, 5), (1, 6, 312 << 10, 8),
    ]
    if len(sys.argv) > 3:
        shapes = [shapes[0]] + shapes[-4:]
    print("%d allocations of 3 x %.2f GiB; columns: GB/s per allocation" % (n_arenas, nbytes / float(1 << 30)))
    for ns, units, slice_b, wps in shapes:
        row = []
        for a in arenas:
            p = a.data_ptr()
            g = ctypes.c_double(0)
            rc = lib.stream3_run(p, p + nbytes, p + 2 * nbytes, ns, nbytes, units, slice_b, wps, ctr.data_ptr(), out.data_ptr(), 3, ctypes.byref(g))
            row.append(g.value if rc == 0 else float("nan"))
        lo, hi = min(row), max(row)
        print("streams %d  unit %d KB  slice %5d KB  waves/SIMD %d :  %s   spread %.1f %%" % (
            ns, units, slice_b >> 10, wps, "  ".join("%6.0f" % x for x in row), 100.0 * (hi - lo) / hi), flush=True)


if __name__ == "__main__":
    main()
