#!/bin/bash
# GPU box, one call: where k_stream's time goes on several workloads -- per-phase clocks (timing build), the ablation ladder, table size.
# usage: tools/r04_diag.sh "<workloads>" "<ablate levels>"
WLS=${1:-"c3 c3r"}; LV=${2:-"1 2 64 4 0"}
O=gpurun_out/r04_diag; mkdir -p $O
for w in $WLS; do
  echo "== $w: product build" | tee -a $O/summary.txt
  bash tools/sweep.sh "$w" "libecb.so" 6 1 2>&1 | tee -a $O/summary.txt
  echo "== $w: per-phase clocks (timing build; it spills: a breakdown, not the kernel's speed)" | tee -a $O/summary.txt
  ECB_LIB=libecb_timing.so ECB_NO_VERIFY=1 timeout -k 10 200 python bench.py --workload $w --steps 2 --warmup 0 --no-cpu-baseline 2>&1 >/dev/null | grep "ecb timing" | tail -1 | tee -a $O/summary.txt
  echo "== $w: ablation ladder" | tee -a $O/summary.txt
  bash tools/tools_ablate.sh $w "$LV" 2>&1 | tee -a $O/summary.txt
done
