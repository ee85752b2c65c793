#!/bin/bash
# usage (GPU box, repo root): tools/diag/ab_trees_bench.sh <tree> ... — bench.py (value, one frame at a time, stage us) per tree
# ('.' or a checkout under .ab/, see ab_trees.sh), three alternating rounds, every run bounded
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD
for i in 1 2 3; do for t in "$@"; do
  echo -n "$t: "
  (cd $R/$t && timeout -k 10 400 python bench.py --cpu-seconds 0 --steps 100 --no-dp-reference --no-opaque --no-coherent --no-runtime-defaults --no-config5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['one_frame_at_a_time']['value'], d['stage_us'])")
done; done
