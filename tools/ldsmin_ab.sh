#!/bin/bash
# From how many frames per run on is the clip state worth keeping in LDS (VBT_SEQ_LDS_MIN)?  Batch 1 / 8 with deferred tracker steps (runs of 4),
# the 34-clip corpus on one GPU (runs of 1-4) and one clip (runs of 64).
cd "${GRAFT_REPO_ROOT:-.}"
for lm in 6 1 2 6 1; do
  for nb in 1 8; do VBT_SEQ_LDS_MIN=$lm DEPTHS=4 timeout -k 10 200 python3 tools/b1_probe.py $nb 2>/dev/null | grep '"track": true' | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('lds_min $lm batch', j['batch'], round(j['frames_per_s']))"; done
  VBT_SEQ_LDS_MIN=$lm timeout -k 10 200 python3 tools/timebatch_bench.py 2>/dev/null | tail -2 | python3 -c "
import json,sys
for ln in sys.stdin:
    j=json.loads(ln); k=list(j)[0]; print('lds_min $lm', k, round(j[k]['frames_per_s']))"
done
