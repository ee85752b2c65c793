#!/bin/bash
# usage (GPU box, repo root): tools/diag/ab_phases.sh <trace lib name in .ab> ...  — per-phase cycles of the blend backward
# (tools/diag/bwd_phases.py) for each -DFR_DIAG_TRACE build named
cd "${GRAFT_REPO_ROOT:-.}"
for v in "$@"; do
  echo "== $v"; FR_HIP_LIB=$PWD/.ab/libfr_$v.so python tools/diag/bwd_phases.py 2>&1 | tail -14
done
