#!/usr/bin/env python
"""GPU box: tools/scale_model.py for N = 8 only (under rocprofv3 via tools/prof_any.sh: which kernels the N-GPU pieces spend their time in)."""
import os, runpy, sys
sys.argv = [sys.argv[0], sys.argv[1] if len(sys.argv) > 1 else "c3", "8", "ranges"]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "scale_model.py"), run_name="__main__")
