#!/bin/bash
# A/B on ONE box: the contract run (bench.py --contract-only, K = 20 and K = 400) of the tree in tools/scratch/old (a worktree of an
# earlier commit with its own built library) against this tree, alternating, three rounds each.  gpurun_out/<tag>/ab.txt
# Set-up (once, on the build host; tools/scratch/ is git-ignored but travels with gpurun):
#   git worktree add -f tools/scratch/old <commit> && (cd tools/scratch/old && python -m vbt_amd.build && make -s -C oracle)
TAG=${1:-ab}
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export VBT_PLAN_FILE=$PWD/profiles/plan_lite0
for K in 20 400; do
  for r in 1 2 3; do
    for arm in old new; do
      if [ $arm = old ]; then D=$PWD/tools/scratch/old; else D=$PWD; fi
      v=$(cd $D && python3 bench.py --steps $K --warmup 5 --contract-only 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.0f %.4f enq %.2f close %.2f' % (j['value'], j['ms_per_step'], j['timed_region_ms']['enqueue'], j['timed_region_ms']['clip_close']))")
      echo "K=$K round $r $arm: $v" | tee -a $OUT/ab.txt
    done
  done
done
