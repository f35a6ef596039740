#!/bin/bash
# GPU box: test the default build, then bench kernel variants back to back.  usage: tools/ab.sh "<libs>" "<workloads>" [steps]
LIBS=${1:-libecb.so}; WLS=${2:-c3}; STEPS=${3:-5}
O=gpurun_out; mkdir -p $O
timeout -k 10 420 python -m pytest tests -m gpu -x -q > $O/ab_test.log 2>&1; rc=$?
tail -5 $O/ab_test.log
[ $rc -ne 0 ] && { echo "tests failed rc=$rc"; grep -n "Error\|error\|assert" $O/ab_test.log | head -30; exit 1; }
for w in $WLS; do for l in $LIBS; do
  ECB_LIB=$l timeout -k 10 200 python bench.py --workload $w --steps $STEPS --no-cpu-baseline 2>$O/ab_err.log | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); r = d['roofline']
        print('%-14s %-3s step %.2f ms  k_stream %.3f ms  %.0f GB/s  frac %.3f  ecs %s exact %s' % ('$l', '$w', d['ms_per_step'], r['kernel_ms_per_launch'], r['achieved'], r['frac'], d['config']['ecs'], d['config']['exactness_pass']))
" || { echo "$l $w failed"; tail -5 $O/ab_err.log; }
done; done
