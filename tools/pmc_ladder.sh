#!/bin/bash
# (needs the profiling build: python -m alntools_amd.build --ablate  ->  alntools_amd/libecb_ablate.so)
# GPU box: VALU / SALU / LDS instruction counts of k_stream per ablation level.  usage: tools/pmc_ladder.sh <workload> "<ablate values>"
W=${1:-c3}; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp ECB_NO_VERIFY=1
for a in ${2:-1 2 4 8 0}; do
  O=/tmp/pmcl_$a; rm -rf $O; mkdir -p $O
  ECB_LIB=libecb_ablate.so ECB_ABLATE=$a timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-include-regex "k_stream<false, false>" --output-format csv -d $O -- python $R/bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline > $O/log 2>&1
  f=$(find $O -name "*counter_collection.csv" | head -1)
  python - "$f" $a <<'PY'
import csv, sys, collections
d = collections.defaultdict(float); n = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_stream' in r.get('Kernel_Name', ''):
        d[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
print("ablate=%-3s" % sys.argv[2], "  ".join("%s %.3fG" % (k.replace('SQ_INSTS_', ''), d[k] / max(n[k], 1) / 1e9) for k in sorted(d)))
PY
done
