#!/bin/bash
# usage (GPU box, repo root): tools/diag/sweep_bwd.sh VAR v1 v2 ...  — isolated blend backward launch (tools/diag/bwd_time.py,
# dispatch-tied events) at config 2, config 5 and the opaque scene with VAR=v, one line each
cd "${GRAFT_REPO_ROOT:-.}"
var=$1; shift
for v in "$@"; do
  for sc in "" "--P 500000 --res 1024 --iters 60" "--opacity 0.9"; do
    env $var=$v python tools/diag/bwd_time.py $sc 2>/dev/null | sed -e "s/^/$var=$v  /" | grep -o "^$var=[^ ]*\|P=[0-9]* res=[0-9]* opacity=[0-9.]*\|'blend_bwd': [0-9.]*\|'blend_fwd': [0-9.]*" | tr '\n' ' '; echo
  done
done
