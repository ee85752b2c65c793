#!/bin/bash
# usage (GPU box, repo root): tools/diag/ab_trees.sh <tree> ...   ('.' = this tree, or a checkout of another commit under .ab/,
# built there: `git archive <commit> | tar -x -C .ab/base && make -C .ab/base/fateavatar_amd/csrc`) — A/B across changes of the
# C ABI, where swapping the library under one Python tree (ab_bench_libs.sh) is not possible.  Per tree: rocprofv3 kernel
# durations of tools/probe.py at config 2, config 5 and the opaque scene, then bench.py twice, alternating; every run bounded.
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD
for t in "$@"; do
  D=$R/$t; tag=$(echo $t | tr '/.' '__')
  for sc in "c2:" "c5:--P 500000 --res 1024 --iters 20" "opaque:--opacity 0.9 --iters 30"; do
    name=${sc%%:*}; args=${sc#*:}
    out=$R/gpurun_out/abt_${tag}_$name; mkdir -p $out
    (cd /tmp && TMPDIR=/tmp timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $out -o p -- python $D/tools/probe.py $args > $out/run.log 2>&1)
    python $R/tools/kstats.py $out/p_results.db > $out/kernels.txt 2>&1
    echo "== $name $t"; grep -E "fr" $out/kernels.txt | awk '{printf "   %-60s calls %s avg %s min %s\n", substr($1,1,60), $2, $3, $4}'
  done
done
for i in 1 2; do for t in "$@"; do
  echo -n "$t: "
  (cd $R/$t && timeout -k 10 400 python bench.py --cpu-seconds 0 --steps 100 --no-dp-reference --no-opaque --no-coherent --no-runtime-defaults --no-config5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['one_frame_at_a_time']['value'], d['stage_us'])")
done; done
