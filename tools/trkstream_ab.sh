#!/bin/bash
# OC-SORT step at the end of the forward's own stream (inline, the default from depth 3) against a tracker stream of its own, 64 clips per step.
cd "${GRAFT_REPO_ROOT:-.}"
for m in inline own inline own; do VBT_TRACKER_STREAM=$m VBT_STRICT_PLACEMENT=0 python3 bench.py --steps 600 --cpu-frames 0 --no-roofline --no-configs --settle-steps 0 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=j['h2d_inclusive']
print('tracker stream $m', round(j['value']), round(h['frames_per_s']))"; done
for m in inline own inline own; do VBT_TRACKER_STREAM=$m VBT_STRICT_PLACEMENT=0 python3 bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-roofline --no-configs 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=j['h2d_inclusive']
print('tracker stream $m K=20', round(j['value']), round(j['value_settled']), round(h['frames_per_s']))"; done
