#!/bin/bash
# A/B two builds of the library on ONE box: .ab/libfr_hip_prev.so vs .ab/libfr_hip_new.so (alternating, 3 rounds)
cd "${GRAFT_REPO_ROOT:-.}"
for r in 1 2 3; do
  for v in prev new; do
    cp .ab/libfr_hip_$v.so fateavatar_amd/libfr_hip.so
    python bench.py --cpu-seconds 0 --steps 200 $@ 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['stage_us'])"
  done
done
cp .ab/libfr_hip_new.so fateavatar_amd/libfr_hip.so
