#!/bin/bash
# usage (GPU box, repo root): tools/diag/sweep_probe.sh "<probe.py args>" VAR v1 v2 ...  — tools/probe.py (forward / backward
# of one frame, event-timed) with VAR=v, one line each
cd "${GRAFT_REPO_ROOT:-.}"
args=$1; var=$2; shift 2
for v in "$@"; do
  echo -n "$var=$v $args: "; env $var=$v python tools/probe.py $args 2>/dev/null | grep "^fwd" 
done
