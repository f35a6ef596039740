#!/bin/bash
# GPU box: instruction counts of k_stream (one PMC pass).  usage: tools/pmc_quick.sh <workload>
W=${1:-c3}; R=$GRAFT_REPO_ROOT; O=/tmp/pmcq; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp ECB_NO_VERIFY=1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --kernel-include-regex "k_stream<false>" --output-format csv -d $O -- python $R/bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline > $O/log 2>&1
f=$(find $O -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
d = collections.defaultdict(float); n = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_stream' in r.get('Kernel_Name', ''):
        d[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for k in sorted(d): print("%-24s %16.0f" % (k, d[k] / max(n[k], 1)))
PY
