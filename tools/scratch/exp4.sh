cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PROBE_K=12 PROBE_REPS=1
for cfg in "4 3 detect" "4 4 detect" "5 3 detect" "5 4 detect" "8 3 detect" "8 4 detect" "8 4 inline" "4 4 inline" "6 4 detect" "7 4 detect"; do
set -- $cfg
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/qmap/q$1d$2$3 -o t -- python3 tools/scratch/depth_probe3.py $1 $2 $3 > /dev/null 2>&1 || exit 1
done
ls gpurun_out/qmap
