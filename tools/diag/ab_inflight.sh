#!/bin/bash
# usage (GPU box, repo root): tools/diag/ab_inflight.sh <tree> ... — rocprofv3 durations of the batched kernels with three chains of four
# views in flight (bench.py --chains ${CHAINS:-4x3}), per tree ('.' or a checkout under .ab/), and the run's own value; every run bounded
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD
for t in "$@"; do
  tag=$(echo $t | tr '/.' '__'); out=$R/gpurun_out/inf_$tag; mkdir -p $out
  (cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out -o p -- python $R/$t/bench.py --chains ${CHAINS:-4x3} --steps 60 --warmup 10 --no-dp-reference --cpu-seconds 0 --no-opaque --no-coherent --no-runtime-defaults --no-config5 > $out/run.log 2>&1)
  python $R/tools/kstats.py $out/p_results.db > $out/kernels.txt 2>&1
  echo "== $t  $(tail -1 $out/run.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])" 2>/dev/null)"
  grep -E "fr.*batch" $out/kernels.txt | awk '{printf "   %-60s calls %s avg %s min %s max %s\n", substr($1,1,60), $2, $3, $4, $5}'
done
