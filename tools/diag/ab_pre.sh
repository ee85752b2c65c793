#!/bin/bash
# usage (GPU box, repo root): tools/diag/ab_pre.sh <lib name in .ab or 'hip'> ...  — rocprofv3 durations of the per-Gaussian
# kernels (and the binning / sort ones) per library at config 2 and config 5 (tools/probe.py), one line each
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD
for v in "$@"; do
  L=$R/.ab/libfr_$v.so; [ $v = hip ] && L=$R/fateavatar_amd/libfr_hip.so
  for sc in "c2:" "c5:--P 500000 --res 1024 --iters 20"; do
    name=${sc%%:*}; args=${sc#*:}
    FR_HIP_LIB=$L tools/profile.sh abp_${v}_$name python $R/tools/probe.py $args > /dev/null 2>&1
    echo "== $name $v"; grep -E "preprocess|totals|tile_sort" gpurun_out/abp_${v}_$name/kernels.txt | awk '{printf "   %-60s calls %s avg %s min %s\n", substr($1,1,60), $2, $3, $4}'
  done
done
