#!/bin/bash
# usage (GPU box, repo root): tools/diag/ab_prof5.sh <lib name in .ab or 'hip'>[:HEAVY_PAIRS] ...  — as ab_prof.sh, at BASELINE config 5
for spec in "$@"; do
  v=${spec%%:*}; hp=${spec#*:}; [ "$hp" = "$spec" ] && hp=641
  L=$GRAFT_REPO_ROOT/.ab/libfr_$v.so; [ $v = hip ] && L=$GRAFT_REPO_ROOT/fateavatar_amd/libfr_hip.so
  FR_HEAVY_PAIRS=$hp FR_HIP_LIB=$L tools/profile.sh ab5_$v python $GRAFT_REPO_ROOT/tools/probe.py --P 500000 --res 1024 --iters 20 > /dev/null 2>&1
  echo "== $spec"; grep -E "blend" gpurun_out/ab5_$v/kernels.txt | awk '{printf "   %-60s calls %s avg %s min %s\n", substr($1,1,60), $2, $3, $4}'
done
