#!/bin/bash
# GPU box: tools_prof.sh with environment settings for the profiled run.  usage: tools_prof_env.sh "<VAR=value ...>" <workload> <outdir>
export $1; shift
exec bash $GRAFT_REPO_ROOT/tools/tools_prof.sh "$@"
