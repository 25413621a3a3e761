#!/bin/bash
# A/B on one box: tracker steps of a 64-clip batch one launch per step (default) vs deferred in groups of `depth` steps (VBT_TRACKER_DEFER=1)
TAG=${1:-defer64}
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export VBT_PLAN_FILE=$PWD/profiles/plan_lite0
for K in 20 400; do
  for r in 1 2 3; do
    for arm in 0 1; do
      v=$(VBT_TRACKER_DEFER=$arm python3 bench.py --steps $K --warmup 5 --contract-only 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.0f %.4f' % (j['value'], j['ms_per_step']))")
      echo "K=$K round $r defer=$arm: $v" | tee -a $OUT/ab.txt
    done
  done
done
