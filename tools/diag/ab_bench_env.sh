#!/bin/bash
# usage (GPU box, repo root): tools/diag/ab_bench_env.sh VAR v1 v2 ... — bench.py (value, one frame at a time, stage us) with VAR=v,
# two alternating rounds
cd "${GRAFT_REPO_ROOT:-.}"
var=$1; shift
for i in 1 2; do for v in "$@"; do
  echo -n "$var=$v: "
  env $var=$v python bench.py --cpu-seconds 0 --steps 100 --no-dp-reference --no-opaque --no-coherent --no-runtime-defaults --no-config5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['one_frame_at_a_time']['value'], d['stage_us'])"
done; done
