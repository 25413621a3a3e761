#!/bin/bash
# A/B of the K-biased requantisation (Rq::kb) on one box: the same bench run with and without VBT_NO_KBIAS, alternating.
cd "${GRAFT_REPO_ROOT:-.}"
OUT=${1:-gpurun_out/kb_ab}
mkdir -p $OUT
B="bench.py --cpu-frames 0 --no-roofline --no-extras --settle-steps 0 --steps ${STEPS:-600}"
for i in 1 2 3; do
  python3 $B > $OUT/on_$i.json 2> $OUT/on_$i.err
  VBT_NO_KBIAS=1 python3 $B > $OUT/off_$i.json 2> $OUT/off_$i.err
done
python3 - <<PY
import json,glob
for k in ("on","off"):
    v=[json.loads(open(f).read().strip().splitlines()[-1])["value"] for f in sorted(glob.glob("$OUT/%s_*.json"%k))]
    print(k, [round(x) for x in v])
PY
