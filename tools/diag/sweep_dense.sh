#!/bin/bash
# The bench scene, config 5 and two dense scenes under the per-unit all-pairs thresholds (FR_DENSE_PAIRS_FWD / _BWD;
# 9999 = never, 0 = always).
cd "${GRAFT_REPO_ROOT:-.}"
if [ "$1" = hist ]; then
  FR_DEBUG_PAIR_HIST=1 python tools/probe.py 2>/dev/null | grep -E "^units|^P="
  FR_DEBUG_PAIR_HIST=1 python tools/probe.py --P 100000 --res 512 --scale 3e-3 --opacity 0.3 2>/dev/null | grep -E "^units|^P="
  FR_DEBUG_PAIR_HIST=1 python tools/probe.py --P 500000 --res 1024 --scale 6.085e-4 --opacity 0.5 2>/dev/null | grep -E "^units|^P="
fi
run() {
  for A in "" "--P 500000 --res 1024" "--P 100000 --res 512 --scale 3e-3 --opacity 0.3" "--P 500000 --res 1024 --scale 6.085e-4 --opacity 0.5" "--opacity 0.9"; do
    python bench.py --cpu-seconds 0 --steps 100 $A 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["stage_us"])'
  done
}
echo "== default"; run
[ "$1" = default ] && exit 0
echo "== never all-pairs"; FR_DENSE_PAIRS_FWD=9999 FR_DENSE_PAIRS_BWD=9999 run
echo "== always all-pairs"; FR_DENSE_PAIRS_FWD=0 FR_DENSE_PAIRS_BWD=0 run
