#!/usr/bin/env python
"""Config 3's k_stream time against WHERE its buffers happen to sit: one process, the same tuples, and (i) a fresh handle (EC table,
key arena, per-read slot ids: new allocations) per trial, (ii) the tuple streams copied to new allocations per trial.  The pool's
"slow" and "fast" boxes turned out to be slow and fast RUNS on one box (profiles/r04_ab_4way.txt): this says which allocation it is."""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from alntools_amd import ecb  # noqa: E402


def timed(b, rid, loc, hf, steps=3):
    abl = bool(os.environ.get("ECB_ABLATE"))        # (profiling build with phases switched off: no ECs, nothing to finalize)

    def fin():
        if not abl:
            b.finalize()
    b.reset(); b.push_device(rid, loc, hf); fin()
    b.profile(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        b.reset(); b.push_device(rid, loc, hf); fin()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ms, n, _ = b.profile_read()
    b.profile(False)
    return ms / max(n, 1), dt * 1e3


def read_rate(t3):
    """GB/s of a pure 16-byte-per-lane read sweep over the three stream buffers where they sit (tools/micro/copy_peak.hip)."""
    import ctypes
    lib = os.path.join(ROOT, "tools", "micro", "bin", "libcopy_peak.so")
    if not os.path.exists(lib):
        return None
    L = ctypes.CDLL(lib)
    L.read_peak_run.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    out = []
    dst = torch.empty(1 << 20, dtype=torch.int32, device=t3[0].device)
    for t in t3:
        g = ctypes.c_double(0)
        torch.cuda.synchronize()
        nbytes = (t.numel() * 4) // (1 << 20) * (1 << 20)
        if L.read_peak_run(t.data_ptr(), dst.data_ptr(), nbytes, 3, ctypes.byref(g)) != 0:
            return None
        out.append(round(g.value))
    return out


def ladder(b, rid, loc, hf):
    """k_stream with phases switched off (profiling build only: ECB_LIB=libecb_ablate.so; ECB_ABLATE is read per batch)."""
    out = {}
    for lv in ("1", "64", "0"):
        os.environ["ECB_ABLATE"] = lv
        try:
            b.reset(); b.push_device(rid, loc, hf)
            b.profile(True)
            for _ in range(3):
                b.reset(); b.push_device(rid, loc, hf)
            torch.cuda.synchronize()
            ms, n, _ = b.profile_read()
            b.profile(False)
            out[lv] = round(ms / max(n, 1), 3)
        finally:
            os.environ.pop("ECB_ABLATE", None)
    b.reset()
    return out


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
    trials = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    dev = torch.device("cuda:0")
    R, T, H, paired, _ = bench.WORKLOADS[wl]
    spec = bench.workload_spec(wl)
    rid, loc, hf, st = bench.generate_shard(spec, 0, R, dev)
    cap = 1 << int(os.environ.get("ECB_EC_CAP_LOG2", "24"))
    print("workload %s: %d records; pointers of the streams: %x %x %x" % (wl, st["records"], rid.data_ptr(), loc.data_ptr(), hf.data_ptr()), flush=True)
    print("-- (i) a fresh handle per trial, the streams where they are")
    for t in range(trials):
        with ecb.EcBuilder(T, H, device=0, ec_capacity=cap, arena_capacity=1 << 26) as b:
            b.hint_reads(st["reads"])
            k, s = timed(b, rid, loc, hf)
            print("trial %d  k_stream %.3f ms  step %.2f ms" % (t, k, s), flush=True)
    print("-- (ii) one handle, the streams copied to new allocations per trial (the old ones kept until the new ones exist)")
    with ecb.EcBuilder(T, H, device=0, ec_capacity=cap, arena_capacity=1 << 26) as b:
        b.hint_reads(st["reads"])
        for t in range(trials):
            k, s = timed(b, rid, loc, hf)
            extra = ""
            if "ablate" in ecb.LIB_PATH:
                extra = "   phases (a) / all but the table's memory / whole: %s   read sweep of the three buffers GB/s: %s" % (ladder(b, rid, loc, hf), read_rate((rid, loc, hf)))
            print("trial %d  k_stream %.3f ms  step %.2f ms   streams at %x %x %x%s" % (t, k, s, rid.data_ptr(), loc.data_ptr(), hf.data_ptr(), extra), flush=True)
            r2, l2, h2 = rid.clone(), loc.clone(), hf.clone()
            rid, loc, hf = r2, l2, h2
            torch.cuda.synchronize()
    print("-- (iii) a fresh handle again, after all that")
    for t in range(2):
        with ecb.EcBuilder(T, H, device=0, ec_capacity=cap, arena_capacity=1 << 26) as b:
            b.hint_reads(st["reads"])
            k, s = timed(b, rid, loc, hf)
            print("trial %d  k_stream %.3f ms  step %.2f ms" % (t, k, s), flush=True)


if __name__ == "__main__":
    main()
