#!/bin/bash
# (needs the profiling build: python -m alntools_amd.build --ablate  ->  alntools_amd/libecb_ablate.so)
# GPU box: HBM traffic of k_stream with the EC table switched off (ECB_ABLATE=4) next to the full kernel: where the bytes above
# the algorithmic 12 B/record come from.  usage: tools_pmc_ablate.sh <workload>
W=${1:-c3}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_abl; mkdir -p $O
cd /tmp && export TMPDIR=/tmp ECB_NO_VERIFY=1
for abl in 0 4; do for c in FETCH_SIZE WRITE_SIZE; do
  ECB_LIB=libecb_ablate.so ECB_ABLATE=$abl timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "k_stream<false, false>" --output-format csv -d $O/a${abl}_$c -- python $R/bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline > $O/a${abl}_$c.log 2>&1
  f=$(find $O/a${abl}_$c -name "*counter_collection.csv" | head -1)
  python - "$f" "$abl" "$c" <<'PY'
import csv, sys
v = [float(r['Counter_Value']) for r in csv.DictReader(open(sys.argv[1])) if 'k_stream' in r.get('Kernel_Name', '') and r['Counter_Name'] == sys.argv[3]]
print("ECB_ABLATE=%s %-10s %14.0f KiB (%d dispatch)" % (sys.argv[2], sys.argv[3], sum(v), len(v)))
PY
  rm -rf $O/a${abl}_$c
done; done
