#!/bin/bash
# Forward of the 64-clip step as eager launches (default above batch 8) or as a replayed hipGraph (VBT_GRAPH_MAX_BATCH=64): bench.py at 600 steps and at K = 20.
cd "${GRAFT_REPO_ROOT:-.}"
for g in 8 64 8 64; do VBT_GRAPH_MAX_BATCH=$g python3 bench.py --steps 600 --cpu-frames 0 --no-roofline --no-configs --settle-steps 0 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=j['h2d_inclusive']
print('graph_max $g  600 steps:', round(j['value']), round(h['frames_per_s']))"; done
for g in 8 64 8 64; do VBT_GRAPH_MAX_BATCH=$g python3 bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-roofline --no-configs 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=j['h2d_inclusive']
print('graph_max $g  K=20:', round(j['value']), round(j['value_settled']), round(h['frames_per_s']))"; done
