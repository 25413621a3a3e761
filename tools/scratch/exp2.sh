cd $GRAFT_REPO_ROOT
export VBT_PLAN_FILE=/tmp/plan_q
VBT_AUTOTUNE_VERBOSE=2 python tools/step_times.py 64 > gpurun_out/steps_e2.txt 2>&1
grep "fused_mbconv v" gpurun_out/steps_e2.txt | grep -E "op (5|9|12|16|20) "
grep -E "total" gpurun_out/steps_e2.txt
cp /tmp/plan_q.b64.f0 gpurun_out/plan_q.b64.f0
diff gpurun_out/plan_q.b64.f0 profiles/plan_lite0.b64.f0
python bench.py --steps 300 --warmup 10 --cpu-frames 0 --no-extras --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('depth3', round(d['value']), round(d['ms_per_step'],4))"
