#!/bin/bash
# GPU box: bench one workload under several values of one env var: tools_env.sh <workload> <VAR> v1 v2 ...
W=$1; V=$2; shift 2
for round in 1 2; do for x in "$@"; do
  env $V=$x timeout -k 5 120 python bench.py --workload $W --steps 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$V=%-6s ms_per_step=%.2f k_stream_ms=%.3f GB/s=%.0f' % ('$x', d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['roofline']['achieved']))
"
done; done
