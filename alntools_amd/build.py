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
    newest = max(os.path.getmtime(p) for p in (SRC, os.path.join(HERE, "..", "include", "ecb.h")))
    return os.path.getmtime(OUT) < newest


def build(force=False, verbose=False):
    """Compile the HIP extension; returns the path of the shared library."""
    if not force and not needs_build():
        return OUT
    cmd = [HIPCC] + FLAGS + ["-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
