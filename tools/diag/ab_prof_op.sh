#!/bin/bash
# usage (GPU box, repo root): tools/diag/ab_prof_op.sh <opacity> <lib name in .ab or 'hip'>[:HEAVY_PAIRS] ...  — as ab_prof.sh, config 2 with another opacity
op=$1; shift
for spec in "$@"; do
  v=${spec%%:*}; hp=${spec#*:}; [ "$hp" = "$spec" ] && hp=641
  L=$GRAFT_REPO_ROOT/.ab/libfr_$v.so; [ $v = hip ] && L=$GRAFT_REPO_ROOT/fateavatar_amd/libfr_hip.so
  FR_HEAVY_PAIRS=$hp FR_HIP_LIB=$L tools/profile.sh abo_$v python $GRAFT_REPO_ROOT/tools/probe.py --opacity $op > /dev/null 2>&1
  echo "== opacity $op $spec"; grep -E "blend_bwd" gpurun_out/abo_$v/kernels.txt | awk '{printf "   %-60s calls %s avg %s min %s\n", substr($1,1,60), $2, $3, $4}'
done
