#!/bin/bash
# GPU box, one call: config 3 first; when its k_stream reads >= SLOW ms (one of the pool's slow boxes) the ablation ladder and an
# L2 counter pass follow on the same box (profiles/r04_slowbox_*); then the secondary workloads back to back.
# usage: tools/r04_box.sh "<workloads after c3>" [lib] [slow_ms] [force_ladder]
WLS=${1:-"c3r c3q c3s c2 dip"}; LIB=${2:-libecb.so}; SLOW=${3:-9.1}; FORCE=${4:-0}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_box; mkdir -p $O
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('%-22s %-4s step %.2f ms  k_stream %.3f ms  %.0f GB/s  frac %.3f  ecs %s  copy %.0f read %.0f' % ('$1', '$2', d['ms_per_step'], r['kernel_ms_per_launch'], r['achieved'], r['frac'], d['config']['ecs'], r.get('peak_measured_copy') or 0, r.get('peak_measured_read') or 0))
"; }
ECB_LIB=$LIB ECB_NO_VERIFY=1 timeout -k 10 300 python bench.py --workload c3 --steps 8 --no-cpu-baseline 2>$O/c3.err | tee $O/c3.json | line $LIB c3 | tee $O/summary.txt
K=$(python -c "
import json
for l in open('$O/c3.json'):
    if l.startswith('{'): print(json.loads(l)['roofline']['kernel_ms_per_launch'])
")
echo "c3 k_stream = $K ms"
if python -c "import sys; sys.exit(0 if (float('$K') >= float('$SLOW') or '$FORCE' == '1') else 1)"; then
  echo "== ladder on this box (k_stream $K ms) ==" | tee -a $O/summary.txt
  bash tools/tools_ablate.sh c3 "1 2 64 4 0" 2>&1 | tee -a $O/summary.txt
  cd /tmp && export TMPDIR=/tmp
  for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    D=/tmp/r04pmc_$$_${set// /_}; rm -rf $D; mkdir -p $D
    ECB_LIB=$LIB ECB_NO_VERIFY=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "k_stream<false, false>" --output-format csv -d $D -- python $R/bench.py --workload c3 --steps 1 --warmup 0 --no-cpu-baseline > $D/log 2>&1
    f=$(find $D -name "*counter_collection.csv" | head -1)
    python - "$f" <<'PY' | tee -a $O/summary.txt
import csv, sys, collections
d = collections.defaultdict(float); n = collections.defaultdict(int)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if 'k_stream' in r.get('Kernel_Name', ''):
            d[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    for k in sorted(d): print("%-28s %18.0f  (per dispatch, %d dispatches)" % (k, d[k] / max(n[k], 1), n[k]))
except Exception as e:
    print("pmc pass failed:", e)
PY
  done
  cd $R
fi
for w in $WLS; do
  ECB_LIB=$LIB ECB_NO_VERIFY=1 timeout -k 10 300 python bench.py --workload $w --steps 6 --no-cpu-baseline 2>$O/$w.err | tee $O/$w.json | line $LIB $w | tee -a $O/summary.txt
done
ECB_LIB=$LIB ECB_NO_VERIFY=1 timeout -k 10 300 python bench.py --workload c3 --steps 8 --no-cpu-baseline 2>>$O/c3.err | line $LIB c3 | tee -a $O/summary.txt
