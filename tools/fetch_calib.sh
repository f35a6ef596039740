#!/bin/bash
# GPU box: calibrate FETCH_SIZE on known byte counts (tools/micro/fetch_calib.hip).  usage: tools/fetch_calib.sh
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/micro/fetch_calib.hip -o /tmp/fetch_calib 2>/dev/null || exit 1
rm -rf /tmp/fc; timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/fc -- /tmp/fetch_calib > /tmp/fc.log 2>&1
grep "k_calib" /tmp/fc.log
python - "$(find /tmp/fc -name '*counter_collection.csv' | head -1)" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_calib' in r['Kernel_Name']:
        kib = float(r['Counter_Value'])
        known = (8 << 30) if 'stream' in r['Kernel_Name'] else (100 << 20) * 64
        print("%-16s FETCH_SIZE %.0f KiB = %.3f GB reported for %.3f GB read: factor %.3f" % (r['Kernel_Name'].split('(')[0], kib, kib * 1024 / 1e9, known / 1e9, known / (kib * 1024)))
PY
