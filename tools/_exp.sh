set -x
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 300 --warmup 10 --cpu-frames 0 --no-extras --no-roofline"
for d in 3 4; do
  VBT_PIPELINE_DEPTH=$d $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('base depth',$d, round(d['value']), d['ms_per_step'])"
done
for img in 100 400; do for d in 3 4 6; do
  VBT_PLAN_FILE=/tmp/plan_img$img VBT_PREFER_IMAGE=$img VBT_PIPELINE_DEPTH=$d $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('image<=$img depth',$d, round(d['value']), d['ms_per_step'])"
done; done
