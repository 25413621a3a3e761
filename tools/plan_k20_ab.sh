#!/bin/bash
# The plan searched under load against the isolated autotuner's plan at the DRIVER's run shape (--steps 20 --warmup 5), alternating.
cd "${GRAFT_REPO_ROOT:-.}"
OTHER=$1
for i in 1 2 3 4; do
  for p in default $OTHER; do
    if [ "$p" = default ]; then unset VBT_PLAN_FILE; else export VBT_PLAN_FILE=$PWD/$p; fi
    python3 bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('plan $p', round(j['value']), round(j['value_settled']), 'roofline frac', round(r['frac'],3), 'family us', round(r['avg_launch_us'],1))"
  done
done
