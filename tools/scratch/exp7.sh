cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_detector.py -x -q 2>&1 | tail -2
export VBT_PLAN_FILE=/tmp/plan_c
VBT_AUTOTUNE_VERBOSE=1 python tools/step_times.py 64 > gpurun_out/steps_e3.txt 2>&1
grep -E "node_chain|alt1" gpurun_out/steps_e3.txt | cut -c1-330 | head -20
grep -E "total" gpurun_out/steps_e3.txt
cp /tmp/plan_c.b64.f0 gpurun_out/plan_c.b64.f0
python bench.py --steps 300 --warmup 10 --cpu-frames 0 --no-extras --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('depth3 (tuned in process)', round(d['value']), round(d['ms_per_step'],4))"
python bench.py --steps 300 --warmup 10 --cpu-frames 0 --no-extras --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('depth3 (plan from file)', round(d['value']), round(d['ms_per_step'],4))"
