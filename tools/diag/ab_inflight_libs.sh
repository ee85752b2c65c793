#!/bin/bash
# usage (GPU box, repo root): [ENVV="VAR=val ..."] tools/diag/ab_inflight_libs.sh <lib name in .ab or 'hip'> ... — rocprofv3 durations of the batched kernels
# with three chains of four views in flight (bench.py --chains 4x3) and the run's value, per library variant; every run bounded
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD
for v in "$@"; do
  L=$R/.ab/libfr_$v.so; [ $v = hip ] && L=$R/fateavatar_amd/libfr_hip.so
  E=""; [ $v != hip ] && [ $v != base ] && E="$ENVV"
  out=$R/gpurun_out/infl_$v; mkdir -p $out
  (cd /tmp && env $E FR_HIP_LIB=$L TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out -o p -- python $R/bench.py --chains 4x3 --steps 60 --warmup 10 --no-dp-reference --cpu-seconds 0 --no-opaque --no-coherent --no-runtime-defaults --no-config5 > $out/run.log 2>&1)
  python $R/tools/kstats.py $out/p_results.db > $out/kernels.txt 2>&1
  echo "== $v  $(tail -1 $out/run.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['one_frame_at_a_time']['value'])" 2>/dev/null)"
  grep -E "fr.*(batch|sparseENS)" $out/kernels.txt | awk '{printf "   %-60s calls %s avg %s min %s max %s\n", substr($1,1,60), $2, $3, $4, $5}'
done
