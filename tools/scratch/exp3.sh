cd $GRAFT_REPO_ROOT
for q in 4 5 8; do for d in 3 4; do for m in detect own inline; do
timeout -k 10 120 python tools/scratch/depth_probe3.py $q $d $m 2>&1 | grep queues || exit 1
done; done; done
