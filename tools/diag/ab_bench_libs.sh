#!/bin/bash
# usage (GPU box, repo root): tools/diag/ab_bench_libs.sh <lib name in .ab or 'hip'> ... — bench.py (value, one frame at a time,
# stage us) per library, two alternating rounds
cd "${GRAFT_REPO_ROOT:-.}"
for i in 1 2; do for v in "$@"; do
  L=$PWD/.ab/libfr_$v.so; [ $v = hip ] && L=$PWD/fateavatar_amd/libfr_hip.so
  echo -n "$v: "
  FR_HIP_LIB=$L python bench.py --cpu-seconds 0 --steps 100 --no-dp-reference --no-opaque --no-coherent --no-runtime-defaults --no-config5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['one_frame_at_a_time']['value'], d['stage_us'])"
done; done
