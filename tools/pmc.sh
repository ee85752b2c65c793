#!/bin/bash
# usage (GPU box): tools/pmc.sh <name> "<counters>" <command...>   -> gpurun_out/<name>/pmc.txt
set -e
name=$1; shift
ctrs=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --pmc $ctrs --kernel-trace -d $out -o c -- "$@" > $out/run_pmc.log 2>&1 || true
python $GRAFT_REPO_ROOT/tools/pmcstats.py $out/c_results.db > $out/pmc.txt 2>&1 || true
cat $out/pmc.txt
