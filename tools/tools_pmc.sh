#!/bin/bash
# GPU box: PMC counters for k_stream (separate passes, no tracing domains besides kernel-trace).  usage: tools_pmc.sh <workload> <outdir>
W=${1:-c2}; OUT=${2:-gpurun_out/pmc}
R=$GRAFT_REPO_ROOT; mkdir -p $R/$OUT; RAW=/tmp/pmc_raw_$$; mkdir -p $RAW   # (raw rocprofv3 output stays on the box: only summaries travel back)
cd /tmp && export TMPDIR=/tmp ECB_NO_VERIFY=1
run() { # name counters...
  name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "k_stream<false, false>" --output-format csv -d $RAW/$name -- python $R/bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline > $R/$OUT/$name.log 2>&1
  f=$(find $RAW/$name -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
d = collections.defaultdict(float); n = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_stream' in r.get('Kernel_Name', ''):
        d[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for k in sorted(d): print("%-28s %18.0f  (per dispatch, %d dispatches)" % (k, d[k] / max(n[k], 1), n[k]))
PY
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
