#!/bin/bash
# GPU box: instruction counts of the stream kernel per library variant.  usage: tools/r04_pmc_libs.sh "<workloads>" "<libs>"
WLS=${1:-c2}; LIBS=${2:-libecb.so}; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp ECB_NO_VERIFY=1
for w in $WLS; do for l in $LIBS; do
  O=/tmp/pmcv_$$_${w}_${l}; rm -rf $O; mkdir -p $O
  ECB_LIB=$l timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-include-regex "k_stream<false, false>" --output-format csv -d $O -- python $R/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline > $O/log 2>&1
  f=$(find $O -name "*counter_collection.csv" | head -1)
  python - "$f" $w $l <<'PY'
import csv, sys, collections
d = collections.defaultdict(float); n = collections.defaultdict(int)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if 'k_stream' in r.get('Kernel_Name', ''):
            d[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    print("%-4s %-18s" % (sys.argv[2], sys.argv[3]), "  ".join("%s %.3fG" % (k.replace('SQ_INSTS_', '').replace('SQ_', ''), d[k] / max(n[k], 1) / 1e9) for k in sorted(d)))
except Exception as e:
    print(sys.argv[2], sys.argv[3], "pmc failed:", e)
PY
done; done
