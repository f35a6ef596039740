#!/bin/bash
# profiling helper (GPU box): k_stream time with phases ablated (ECB_ABLATE bits: 1 stop after (a), 2 after (b), 4 no upsert)
W=${1:-c2}
for a in ${2:-1 2 4 0}; do
  ECB_ABLATE=$a timeout -k 5 120 python bench.py --workload $W --steps 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('ablate=$a ms_per_step=%.2f k_stream_ms=%.2f GB/s=%.0f' % (d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['roofline']['achieved']))
"
done
