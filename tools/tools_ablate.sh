#!/bin/bash
# (needs the profiling build: python -m alntools_amd.build --ablate  ->  alntools_amd/libecb_ablate.so)
# profiling helper (GPU box): k_stream time with phases ablated (ECB_ABLATE bits: 1 stop after (a), 2 after (b), 4 no table,
# 8 no key compare, 16 first 16 bytes of a slot only, 32 no key stores, 64 no table access); "a:cap" sets ECB_EC_CAP_LOG2 too
W=${1:-c2}
for x in ${2:-1 2 4 0}; do
  a=${x%%:*}; cap=${x#*:}; [ "$cap" = "$x" ] && cap=""
  env ECB_LIB=libecb_ablate.so ECB_ABLATE=$a ${cap:+ECB_EC_CAP_LOG2=$cap} ECB_NO_VERIFY=1 timeout -k 5 120 python bench.py --workload $W --steps 8 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('ablate=%-5s ms_per_step=%.2f k_stream_ms=%.2f GB/s=%.0f' % ('$x', d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['roofline']['achieved']))
"
done
