#!/bin/bash
# A/B of two pinned plans on one box: bench.py's contract run, alternating.  usage: plan_ab.sh <other plan prefix> <outdir>
cd "${GRAFT_REPO_ROOT:-.}"
OTHER=$1
OUT=${2:-gpurun_out/plan_ab}
mkdir -p $OUT
B="bench.py --cpu-frames 0 --no-roofline --no-extras --settle-steps 0 --steps ${STEPS:-600}"
for i in 1 2 3; do
  python3 $B > $OUT/new_$i.json 2> $OUT/new_$i.err
  VBT_PLAN_FILE=$PWD/$OTHER python3 $B > $OUT/old_$i.json 2> $OUT/old_$i.err
done
python3 - <<PY
import json,glob
for k in ("new","old"):
    v=[json.loads(open(f).read().strip().splitlines()[-1])["value"] for f in sorted(glob.glob("$OUT/%s_*.json"%k))]
    print(k, [round(x) for x in v])
PY
