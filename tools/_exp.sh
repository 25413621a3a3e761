cd $GRAFT_REPO_ROOT
export VBT_PLAN_FILE=$PWD/profiles/plan_lite0
python tools/step_times.py 64 2>/dev/null | grep -E "expand_dw|total" | cut -c1-112
python bench.py --steps 300 --warmup 10 --cpu-frames 0 --no-extras --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('depth3', round(d['value']), round(d['ms_per_step'],4))"
