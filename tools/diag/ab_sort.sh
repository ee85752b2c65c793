#!/bin/bash
# usage (GPU box, repo root): tools/diag/ab_sort.sh <lib name in .ab or 'hip'> ... — rocprofv3 durations of the sort and blend kernels per
# library variant (same ABI) at config 2 and config 5, every run bounded
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD
for v in "$@"; do
  L=$R/.ab/libfr_$v.so; [ $v = hip ] && L=$R/fateavatar_amd/libfr_hip.so
  for sc in "c2:" "c5:--P 500000 --res 1024 --iters 20"; do
    name=${sc%%:*}; args=${sc#*:}
    out=$R/gpurun_out/abso_${v}_$name; mkdir -p $out
    (cd /tmp && FR_HIP_LIB=$L TMPDIR=/tmp timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out -o p -- python $R/tools/probe.py $args > $out/run.log 2>&1)
    python $R/tools/kstats.py $out/p_results.db > $out/kernels.txt 2>&1
    echo "== $name $v"; grep -E "sort|blend" $out/kernels.txt | awk '{printf "   %-60s calls %s avg %s min %s\n", substr($1,1,60), $2, $3, $4}'
  done
done
