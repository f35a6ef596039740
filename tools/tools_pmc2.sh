#!/bin/bash
# GPU box: PMC counters of the two k_stream dispatches of tools/exp_hits2.py (cold table, then all hits), per dispatch.
OUT=${1:-gpurun_out/pmc2}
R=$GRAFT_REPO_ROOT; mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "k_stream<false, false>" --output-format csv -d $R/$OUT/$name -- python $R/tools/exp_hits2.py > $R/$OUT/$name.log 2>&1
  f=$(find $R/$OUT/$name -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
d = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_stream' in r.get('Kernel_Name', ''):
        d[r['Counter_Name']][int(r['Dispatch_Id'])] += float(r['Counter_Value'])
for k in sorted(d):
    v = [d[k][i] for i in sorted(d[k])]
    print("%-28s %s" % (k, "  ".join("%16.0f" % x for x in v)) + ("   delta %+.1f %%" % (100 * (v[0] - v[-1]) / max(v[-1], 1)) if len(v) > 1 else ""))
PY
  rm -rf $R/$OUT/$name
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run sq3 SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
