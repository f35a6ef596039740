#!/bin/bash
# GPU box: what bounds the kernels between k_stream and the CSR (the step's ~1.3 ms besides the stream kernel): rocprofv3 counters per kernel,
# separate passes.  usage: r04_fin_pmc.sh <workload> <outdir under gpurun_out>
W=${1:-c3}; OUT=${2:-gpurun_out/fin_pmc}
R=$GRAFT_REPO_ROOT; mkdir -p $R/$OUT; RAW=/tmp/finpmc_raw_$$; mkdir -p $RAW
cd /tmp && export TMPDIR=/tmp ECB_NO_VERIFY=1
RE="k_count_bins|k_part_scatter_staged|k_part_hist|k_emit_small|k_rank|k_clear_slots|k_scan_lb|k_popc"
run() { # name counters...
  name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "$RE" --output-format csv -d $RAW/$name -- python $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $R/$OUT/$name.log 2>&1
  f=$(find $RAW/$name -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, re, sys, collections
d = collections.defaultdict(float); n = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r'k_\w+', r.get('Kernel_Name', '')); k = (m.group(0) if m else '?', r['Counter_Name'])
    d[k] += float(r['Counter_Value']); n[k] += 1
for k in sorted(d): print("%-26s %-24s %16.0f  (per dispatch, %d)" % (k[0][:26], k[1], d[k] / max(n[k], 1), n[k]))
PY
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
run tcc3 TCC_HIT_sum TCC_MISS_sum
rm -rf $RAW
