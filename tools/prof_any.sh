#!/bin/bash
# GPU box: rocprofv3 kernel stats of any driver script, libecb's kernels only.  usage: tools/prof_any.sh <outdir> <script> [args]
OUT=$1; shift; R=$GRAFT_REPO_ROOT; mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT -- python $R/"$@" > $R/$OUT/run.log 2>&1
cd $R; f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'anonymous namespace)::' in r['Name'] and '::k_' in r['Name']]
print("%-34s %6s %10s %10s %10s %12s" % ("kernel", "calls", "avg_ms", "min_ms", "max_ms", "total_ms"))
for r in rows:
    name = r['Name'].split('anonymous namespace)::', 1)[1].split('(')[0]
    print("%-34s %6s %10.3f %10.3f %10.3f %12.3f" % (name[:34], r['Calls'], float(r['AverageNs']) / 1e6, float(r.get('MinNs', 0)) / 1e6, float(r.get('MaxNs', 0)) / 1e6, float(r['TotalDurationNs']) / 1e6))
PY
grep -h "shaped" $OUT/run.log
find $OUT -name "*.csv" -delete
