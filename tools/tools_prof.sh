#!/bin/bash
# GPU box: rocprofv3 kernel trace of one bench run; prints libecb's kernels only.  usage: tools_prof.sh <workload> <outdir>
W=${1:-c2}; OUT=${2:-gpurun_out/prof}
R=$GRAFT_REPO_ROOT; mkdir -p $R/$OUT; RAW=/tmp/prof_raw_$$; mkdir -p $RAW
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -- python $R/bench.py --workload $W --steps 5 --warmup 1 --no-cpu-baseline > $R/$OUT/run.log 2>&1
cd $R
f=$(find $RAW -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if ('anonymous namespace)::' in r['Name'] and '::k_' in r['Name']) or 'fillBuffer' in r['Name']]   # (+ the runtime's fill kernels: hipMemsetAsync in reset)
print("%-34s %6s %10s %10s %10s %12s" % ("kernel", "calls", "avg_ms", "min_ms", "max_ms", "total_ms"))
for r in rows:
    name = r['Name'].split('::')[1].split('(')[0] if '::' in r['Name'] else r['Name']
    name = r['Name'].split('anonymous namespace)::', 1)[1].split('(')[0] if '::k_' in r['Name'] else r['Name'].split('(')[0]      # (ks_std::k_stream<...> keeps its namespace)
    print("%-34s %6s %10.3f %10.3f %10.3f %12.3f" % (name[:34], r['Calls'], float(r['AverageNs']) / 1e6, float(r.get('MinNs', 0)) / 1e6, float(r.get('MaxNs', 0)) / 1e6, float(r['TotalDurationNs']) / 1e6))
PY
# the bench line of THIS (profiled) run, whole: its HIP-event time for k_stream is what the profiler's average above must agree with -- another
# process would have its tuples somewhere else in HBM, which moves the kernel by up to 14 %
grep '^{' $OUT/run.log | tail -1 > $OUT/profiled_run.json
python - $OUT/profiled_run.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read()); r = d["roofline"]
print("the profiled run's own bench line: k_stream %.3f ms per launch by HIP events (%s), step %.3f ms, frac %.3f" % (r["kernel_ms_per_launch"], r["kernel"], d["ms_per_step"], r["frac"]))
PY
