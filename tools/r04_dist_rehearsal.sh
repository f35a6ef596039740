#!/bin/bash
# GPU box (one card): bench.py's multi-rank path rehearsed over gloo with the ranks sharing the card.  usage: r04_dist_rehearsal.sh <outdir under gpurun_out>
R=$GRAFT_REPO_ROOT; O=$R/${1:-gpurun_out/dist_rehearsal}; mkdir -p $O
export ECB_DIST_BACKEND=gloo
P=29551
for cfg in "c4t arrays" "tiny arrays" "tiny tiles"; do
  set -- $cfg
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $P $R/bench.py --gpus 2 --workload $1 --layout $2 --steps 2 --warmup 1 --no-cpu-baseline > $O/$1_$2.out 2> $O/$1_$2.err
  echo "$1 $2 rc=$?"; grep '^{' $O/$1_$2.out | python $R/tools/showbench.py || tail -5 $O/$1_$2.err
  P=$((P+1))
done
