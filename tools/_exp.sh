cd $GRAFT_REPO_ROOT
B="python bench.py --steps 300 --warmup 10 --cpu-frames 0 --no-extras --no-roofline"
for c in 2 3; do
VBT_AUTOTUNE_CONCURRENCY=$c VBT_PLAN_FILE=/tmp/plan_c$c $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('autotune concurrency $c:', round(d['value']), round(d['ms_per_step'],4))"
VBT_PLAN_FILE=/tmp/plan_c$c $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  rerun same plan:', round(d['value']), round(d['ms_per_step'],4))"
cp /tmp/plan_c$c.b64.f0 gpurun_out/plan_c$c.b64.f0
done
VBT_PLAN_FILE=$PWD/profiles/plan_lite0 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pinned plan:', round(d['value']), round(d['ms_per_step'],4))"
