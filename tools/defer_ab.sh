#!/bin/bash
# Deferred tracker steps (one time-batched walk per `depth` steps) at the bench's 64 clips per step, against one tracker launch per step.
cd "${GRAFT_REPO_ROOT:-.}"
for d in 0 1 0 1 0 1; do VBT_TRACKER_DEFER=$d python3 bench.py --steps 600 --cpu-frames 0 --no-roofline --no-configs --settle-steps 0 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=j['h2d_inclusive']
print('defer $d', round(j['value']), round(h['frames_per_s']))"; done
for d in 0 1 0 1; do VBT_TRACKER_DEFER=$d python3 bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-roofline --no-configs 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=j['h2d_inclusive']
print('defer $d K=20', round(j['value']), round(j['value_settled']), round(h['frames_per_s']))"; done
