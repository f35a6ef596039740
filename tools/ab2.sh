#!/bin/bash
# GPU box: ablation ladder for several libs: tools/ab2.sh "<libs>" <workload> "<ablate values>"
for l in $1; do echo "== $l"; ECB_LIB=$l bash tools/tools_ablate.sh $2 "$3"; done
