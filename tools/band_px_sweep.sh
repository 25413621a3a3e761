#!/bin/bash
# Developer probe: small-batch pipeline rate against the band size of the row-band kernels (VBT_BAND_PX) and the concurrency the
# autotuner measures under (VBT_AUTOTUNE_CONCURRENCY); every setting tunes its own plan into gpurun_out/<tag>/.
TAG=${1:-r04e}
mkdir -p gpurun_out/$TAG
for px in 320 128 64; do
  for ac in 1 4; do
    for nb in 1 8; do
      VBT_BAND_PX=$px VBT_AUTOTUNE_CONCURRENCY=$ac VBT_PLAN_FILE=$PWD/gpurun_out/$TAG/plan_px${px}_ac${ac} DEPTHS=4 timeout -k 10 200 python tools/b1_probe.py $nb 2>/dev/null | grep batch | sed "s/^/px $px ac $ac: /"
    done
  done
done
