#!/bin/bash
# usage (GPU box, repo root): tools/diag/ab_prof.sh <lib name in .ab or 'hip'> ...   — rocprofv3 kernel durations of ONE
# eager forward+backward frame loop (tools/probe.py) per library, for A/B comparisons of kernel variants
for v in "$@"; do
  L=$GRAFT_REPO_ROOT/.ab/libfr_$v.so; [ $v = hip ] && L=$GRAFT_REPO_ROOT/fateavatar_amd/libfr_hip.so
  FR_HIP_LIB=$L tools/profile.sh ab_$v python $GRAFT_REPO_ROOT/tools/probe.py > /dev/null 2>&1
  echo "== $v"; grep -E "blend|sort|preprocess|totals" gpurun_out/ab_$v/kernels.txt | awk '{printf "   %-70s calls %s avg %s min %s\n", substr($1,1,70), $2, $3, $4}'
done
