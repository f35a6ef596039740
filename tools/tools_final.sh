#!/bin/bash
# GPU box: everything profiles/ holds for one workload: bench line, rocprofv3 kernel stats, PMC passes, the traffic file bench.py reads.
# usage: tools_final.sh <workload> <round tag> [commit]      (results in gpurun_out/final_<workload>/; commit: the build's, `git rev-parse --short HEAD`
#        where the call is made -- the GPU box has no .git)
W=${1:-c3}; TAG=${2:-r04}; COMMIT=${3:-unknown}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final_$W; mkdir -p $O
timeout -k 10 500 python $R/bench.py --workload $W > $O/bench.log 2>$O/bench.err && grep '^{' $O/bench.log > $O/${TAG}_bench_${W}_n1.json
bash $R/tools/tools_prof.sh $W gpurun_out/final_$W/prof > $O/${TAG}_rocprof_kernel_stats_$W.txt 2>&1; cp $O/prof/profiled_run.json $O/${TAG}_bench_${W}_n1_profiled_run.json 2>/dev/null; rm -rf $O/prof
bash $R/tools/tools_pmc.sh $W gpurun_out/final_$W/pmc > $O/${TAG}_pmc_k_stream_$W.txt 2>&1; rm -rf $O/pmc
# FETCH_SIZE once more with the EC table switched off (ECB_ABLATE=4): what is left is the record streams (16-byte-per-lane streaming
# loads, which gfx950 reports at half their bytes); the difference to the full kernel is the table's 64-byte lines (reported 1:1,
# tools/fetch_calib.sh).  traffic = 2 x stream part + table part + WRITE_SIZE.
cd /tmp && export TMPDIR=/tmp ECB_NO_VERIFY=1
ECB_LIB=libecb_ablate.so ECB_ABLATE=4 timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "k_stream<false, false>" --output-format csv -d $O/abl -- python $R/bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline > $O/abl.log 2>&1
cd $R
python - "$(find $O/abl -name '*counter_collection.csv' | head -1)" $O/${TAG}_pmc_k_stream_$W.txt $W $TAG $COMMIT $O/${TAG}_bench_${W}_n1.json > $O/traffic_${W}_n1.json <<'PY'
import csv, json, re, sys
try:
    kernel = json.loads(open(sys.argv[6]).read().splitlines()[0])["roofline"]["kernel"]      # (the compilation of k_stream.inc that ran: ecb_profile_kernel)
except Exception:
    kernel = "ks_std::k_stream<false, false>"
v = [float(r['Counter_Value']) for r in csv.DictReader(open(sys.argv[1])) if 'k_stream' in r.get('Kernel_Name', '') and r['Counter_Name'] == 'FETCH_SIZE']
stream_kib = sum(v) / max(len(v), 1)
t = open(sys.argv[2]).read()
fetch_kib = float(re.search(r'FETCH_SIZE\s+(\d+)', t).group(1)); write_kib = float(re.search(r'WRITE_SIZE\s+(\d+)', t).group(1))
table_kib = max(fetch_kib - stream_kib, 0.0)
print(json.dumps({"kernel": kernel, "workload": sys.argv[3], "n_gpus": 1, "round": sys.argv[4], "commit": sys.argv[5],
                  "FETCH_SIZE_KiB": fetch_kib, "FETCH_SIZE_KiB_without_ec_table": stream_kib, "WRITE_SIZE_KiB": write_kib,
                  "hbm_bytes_per_launch": int((2 * stream_kib + table_kib + write_kib) * 1024),
                  "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/tools_final.sh); the record streams' share of FETCH_SIZE "
                            "(measured with the EC table switched off) doubled, the table's 64-byte lines and WRITE_SIZE taken 1:1 "
                            "(calibration on known byte counts: profiles/r02_fetch_size_calibration.txt)"}, indent=1))
PY
rm -rf $O/abl
ls -la $O
