#!/bin/bash
# GPU box: everything profiles/ holds for one workload: bench line, rocprofv3 kernel stats, PMC passes.  usage: tools_final.sh <workload> <round tag>
W=${1:-c3}; TAG=${2:-r01}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final_$W; mkdir -p $O
timeout -k 10 400 python $R/bench.py --workload $W > $O/bench.log 2>$O/bench.err && grep '^{' $O/bench.log > $O/${TAG}_bench_${W}_n1.json
bash $R/tools/tools_prof.sh $W gpurun_out/final_$W/prof > $O/${TAG}_rocprof_kernel_stats_$W.txt 2>&1; rm -rf $O/prof
bash $R/tools/tools_pmc.sh $W gpurun_out/final_$W/pmc > $O/${TAG}_pmc_k_stream_$W.txt 2>&1; rm -rf $O/pmc/*/
ls -la $O
