#!/bin/bash
# build kernel variants here (container), run on the GPU box:  tools_variants.sh build "name:-DFLAG=.. ..." ... ; tools_variants.sh run c2 name...
if [ "$1" == "build" ]; then shift
  for spec in "$@"; do name=${spec%%:*}; flags=${spec#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value $flags -o alntools_amd/libecb_$name.so alntools_amd/csrc/ecb.hip || exit 1
  done
else shift; W=$1; shift
  for round in 1 2; do for name in "$@"; do
    ECB_LIB=libecb_$name.so timeout -k 5 120 python bench.py --workload $W --steps 6 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-14s ms_per_step=%.2f k_stream_ms=%.3f GB/s=%.0f' % ('$name', d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['roofline']['achieved']))
"
  done; done
fi
