cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_detector.py -x -q 2>&1 | tail -4
export VBT_PLAN_FILE=/tmp/plan_b1
python tools/step_times.py 64 2>/dev/null | cut -c1-118 > gpurun_out/r2_steps4.txt; grep -E "band|heads|node|total" gpurun_out/r2_steps4.txt | cut -c1-112
python bench.py --steps 300 --warmup 10 --cpu-frames 0 --no-extras --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('depth3', round(d['value']), round(d['ms_per_step'],4))"
cp /tmp/plan_b1.b64.f0 gpurun_out/plan_b1.b64.f0
