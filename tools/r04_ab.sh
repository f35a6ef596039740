#!/bin/bash
# GPU box, one call: a subset of the GPU parity tests on the default build, then kernel variants back to back on several workloads.
# usage: tools/r04_ab.sh "<pytest -k expression or ''>" "<libs>" "<workloads>" [steps] [reps]
KEXPR=$1; LIBS=${2:-libecb.so}; WLS=${3:-c3}; STEPS=${4:-6}; REPS=${5:-1}
O=gpurun_out/r04_ab; mkdir -p $O
if [ -n "$KEXPR" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$KEXPR" > $O/test.log 2>&1; rc=$?
  tail -4 $O/test.log
  [ $rc -ne 0 ] && { echo "tests failed rc=$rc"; grep -n "Error\|error\|assert" $O/test.log | head -30; exit 1; }
fi
bash tools/sweep.sh "$WLS" "$LIBS" $STEPS $REPS 2>&1 | tee $O/sweep.txt
