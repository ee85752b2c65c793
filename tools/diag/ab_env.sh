#!/bin/bash
# usage (GPU box, repo root): tools/diag/ab_env.sh VAR v1 v2 ...  — rocprofv3 kernel durations of tools/probe.py with VAR=v
var=$1; shift
n=0
for v in "$@"; do
  n=$((n+1))
  env $var=$v tools/profile.sh abe_$n python $GRAFT_REPO_ROOT/tools/probe.py > /dev/null 2>&1
  echo "== $var=$v"; grep -E "blend" gpurun_out/abe_$n/kernels.txt | awk '{printf "   %-60s calls %s avg %s min %s\n", substr($1,1,60), $2, $3, $4}'
done
