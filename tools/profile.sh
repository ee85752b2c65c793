#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile.sh <name> <command...>
# writes gpurun_out/<name>/ with the rocprofv3 kernel trace db + a text summary
set -e
name=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $out -o p -- "$@" > $out/run.log 2>&1 || true
python $GRAFT_REPO_ROOT/tools/kstats.py $out/p_results.db > $out/kernels.txt 2>&1 || true
cat $out/kernels.txt
python $GRAFT_REPO_ROOT/tools/ktimeline.py $out/p_results.db > $out/timeline.txt 2>&1 || true
