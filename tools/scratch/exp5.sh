cd $GRAFT_REPO_ROOT
export VBT_PLAN_FILE=$PWD/profiles/plan_lite0
for cfg in "8 own" "4 own" "4 inline" "3 inline" "2 inline"; do
set -- $cfg
GPU_MAX_HW_QUEUES=$1 VBT_TRACKER_STREAM=$2 python bench.py --steps 300 --warmup 10 --cpu-frames 0 --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2', 'fps', round(d['value']), 'ms', round(d['ms_per_step'],4), 'h2d', round(d['value_h2d_inclusive']), 'det', round(d['splits']['detect_only']['frames_per_s']))"
GPU_MAX_HW_QUEUES=$1 VBT_TRACKER_STREAM=$2 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-roofline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2', 'k20 fps', round(d['value']), 'ms', round(d['ms_per_step'],4))"
done
