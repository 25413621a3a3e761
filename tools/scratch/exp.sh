cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_detector.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -2
export VBT_PLAN_FILE=$PWD/profiles/plan_lite0
python tools/step_times.py 64 > gpurun_out/steps_e1.txt 2>&1
grep -E "^ *[0-9]+ (fused_mbconv|fused_stem|fused_sepconv_band)|total" gpurun_out/steps_e1.txt | cut -c1-112
python bench.py --steps 300 --warmup 10 --cpu-frames 0 --no-extras --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('depth3', round(d['value']), round(d['ms_per_step'],4))"
python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('k20', round(d['value']), round(d['ms_per_step'],4), d['timed_region_ms'])"
