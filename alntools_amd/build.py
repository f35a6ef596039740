# -*- coding: utf-8 -*-
"""Build ``libecb.so`` (HIP, gfx950) in-tree with hipcc.  ``python -m alntools_amd.build``."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "ecb.hip")
OUT = os.path.join(HERE, "libecb.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Wno-unused-value",
         "-Wno-unused-function"]


def needs_build():
    if not os.path.exists(OUT):
        return True
    newest = max(os.path.getmtime(p) for p in (SRC, os.path.join(HERE, "csrc", "k_stream.inc"), os.path.join(HERE, "..", "include", "ecb.h")))
    return os.path.getmtime(OUT) < newest


def build(force=False, verbose=False):
    """Compile the HIP extension; returns the path of the shared library."""
    from . import bamdec
    try:
        bamdec.build(force=force)                # the host side's BAM decoder (plain C + zlib): a library of its own, and optional --
    except (OSError, subprocess.CalledProcessError) as e:             # without it the pure-Python reader (or pysam) decodes
        print("warning: libbamdec.so not built (%s): BAM files will be decoded by the Python reader" % e, file=sys.stderr)
    if force or needs_build():
        cmd = [HIPCC] + FLAGS + ["-o", OUT, SRC]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    try:
        build_tools(force=force)                 # (a measurement helper: never the reason the product library is missing)
    except (OSError, subprocess.CalledProcessError) as e:
        print("warning: tools/micro/copy_peak.hip not built (%s): bench.py falls back to torch's copy for its measured peak" % e, file=sys.stderr)
    return OUT


def build_tools(force=False):
    """Measurement helpers that bench.py loads when they are there (never the product): tools/micro/copy_peak.hip."""
    src = os.path.join(HERE, "..", "tools", "micro", "copy_peak.hip")
    out = os.path.join(HERE, "..", "tools", "micro", "bin", "libcopy_peak.so")
    if not os.path.exists(src) or (not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src)):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", "-o", out, src])
    return out


def build_sanitized(out=None, verbose=False):
    """Host side of libecb under AddressSanitizer + UndefinedBehaviorSanitizer (the device code is compiled as usual:
    GPU sanitizers are not available): ``libecb_asan.so``, for the no-GPU argument / state checking tests.  The C++ half of
    the library is ~1.5 kLoC of manual hipMalloc / hipFree pairs and early returns."""
    out = out or os.path.join(HERE, "libecb_asan.so")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-shared", "-fPIC", "-fsanitize=address,undefined",
           "-fno-gpu-sanitize", "-shared-libsan", "-Wno-unused-value", "-o", out, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build_ablate(verbose=False):
    """Profiling build ``libecb_ablate.so`` (-DECB_ABLATE_RT): phases of k_stream can be switched off with env ECB_ABLATE;
    used through ``ECB_LIB=libecb_ablate.so`` by tools/pmc_ladder.sh, tools_ablate.sh and tools_final.sh -- never by the product."""
    out = os.path.join(HERE, "libecb_ablate.so")
    cmd = [HIPCC] + FLAGS + ["-DECB_ABLATE_RT", "-o", out, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    if "--ablate" in sys.argv:
        print(build_ablate(verbose=True))
    elif "--asan" in sys.argv:
        print(build_sanitized(verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
