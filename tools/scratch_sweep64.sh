cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04f
for d in 3 4; do for m in inline; do PROBE_REPS=2 timeout -k 10 120 python tools/depth_probe.py 4 $d $m 2>/dev/null | grep queues; done; done
PROBE_REPS=2 timeout -k 10 120 python tools/depth_probe.py 4 3 detect 2>/dev/null | grep queues
for px in 200 256 400 480; do echo "band px $px (autotuned plan):"; VBT_BAND_PX=$px VBT_PLAN_FILE=$PWD/gpurun_out/r04f/plan_px$px PROBE_REPS=2 timeout -k 10 200 python tools/depth_probe.py 4 3 inline 2>/dev/null | grep queues; done
echo "autotuned fresh plan px 320:"; VBT_PLAN_FILE=$PWD/gpurun_out/r04f/plan_fresh PROBE_REPS=2 timeout -k 10 200 python tools/depth_probe.py 4 3 inline 2>/dev/null | grep queues
