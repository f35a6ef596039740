#!/bin/bash
# GPU box: kernel timeline of one bench step (start offsets and gaps), from rocprofv3 --kernel-trace.  usage: timeline.sh <workload>
W=${1:-c3}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tl_$W; mkdir -p $O
cd /tmp && export TMPDIR=/tmp ECB_NO_VERIFY=1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -- python $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > $O/run.log 2>&1
cd $R
f=$(find $O -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last two steps: from the second-to-last k_stream<false> on (the reset's fills of the last step sit between them)
ks = [i for i, r in enumerate(rows) if 'k_stream' in r['Kernel_Name']]
i0 = ks[-2]
t0 = int(rows[i0]['Start_Timestamp']); prev_end = t0
print("%-28s %10s %10s %10s" % ("kernel", "start_us", "dur_us", "gap_us"))
for r in rows[i0:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = r['Kernel_Name']; n = n[n.index('k_'):].split('(')[0] if 'k_' in n else n[:28]
    print("%-28s %10.1f %10.1f %10.1f" % (n[:28], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = e
print("total from k_stream start to last kernel end: %.1f us" % ((prev_end - t0) / 1e3))
PY
rm -rf $O
