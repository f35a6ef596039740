#!/bin/bash
# GPU box: A/B kernel variants (ECB_LIB) and environment knobs back to back, REPS rounds of all of them in turn (boxes drift: compare within a round).
# usage: tools/sweep.sh "<workloads>" "<lib[:ENV=val[,ENV=val]]> ..." [steps] [reps]
WLS=${1:-c3}; SPECS=${2:-libecb.so}; STEPS=${3:-6}; REPS=${4:-1}
for rep in $(seq $REPS); do for w in $WLS; do for spec in $SPECS; do
  lib=${spec%%:*}; envs=${spec#*:}; [ "$envs" = "$spec" ] && envs=""
  env ECB_LIB=$lib ECB_NO_VERIFY=1 ${envs//,/ } timeout -k 10 200 python bench.py --workload $w --steps $STEPS --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); r = d['roofline']
        print('%-34s %-4s step %.2f ms  k_stream %.3f ms  %.0f GB/s  frac %.3f  ecs %s' % ('$spec', '$w', d['ms_per_step'], r['kernel_ms_per_launch'], r['achieved'], r['frac'], d['config']['ecs']))
" || echo "$spec $w failed"
done; done; done
