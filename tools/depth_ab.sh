#!/bin/bash
# Pipeline depth 3 vs 4 with the pinned plan (resident and host-fed, bench.py at 600 steps), and EfficientDet-Lite2 at depths 2 / 3 / 4.
cd "${GRAFT_REPO_ROOT:-.}"
for d in 3 4 3 4; do VBT_PIPELINE_DEPTH=$d python3 bench.py --steps 600 --cpu-frames 0 --no-roofline --no-configs --settle-steps 0 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=j['h2d_inclusive']
print('depth $d', round(j['value']), round(h['frames_per_s']))"; done
for d in 2 3 4; do python3 tools/lite2_probe.py $d 2>/dev/null | tail -1; done
