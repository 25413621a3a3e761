#!/bin/bash
# One measurement session on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_session.sh r03
# writes raw outputs under gpurun_out/<tag>/ ; tools/make_profile_summary.py <tag> turns them into profiles/<tag>_*.
# Counter passes are separate rocprofv3 runs with --kernel-trace only (no other trace domain), depth 1, pinned plan.
set -u
TAG=${1:-r03}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
export VBT_PLAN_FILE=$PWD/profiles/plan_lite0
OUT=gpurun_out/$TAG
mkdir -p $OUT
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench default done"
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err
echo "bench k20 done"
session() {   # $1 = output prefix ("" | lite2_), environment selects model and plan
  local PRE=$1
  export VBT_STRICT_PLACEMENT=0   # under rocprofv3 the placement probe may read every pair of streams as serialised (exported: the program after `--` must be python3 itself)
  local B="bench.py --steps 50 --warmup 5 --cpu-frames 0 --no-extras --no-roofline --settle-steps 0"
  VBT_PIPELINE_DEPTH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${PRE}trace_d1 -o d1 -- python3 $B > /dev/null 2> $OUT/${PRE}trace_d1.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${PRE}trace_d3 -o d3 -- python3 $B > /dev/null 2> $OUT/${PRE}trace_d3.err
  echo "${PRE}traces done"
  local P="bench.py --steps 6 --warmup 2 --cpu-frames 0 --no-extras --no-roofline --settle-steps 0"
  export VBT_PIPELINE_DEPTH=1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${PRE}pmc_fetch -o f -- python3 $P > /dev/null 2> $OUT/${PRE}pmc_fetch.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${PRE}pmc_write -o w -- python3 $P > /dev/null 2> $OUT/${PRE}pmc_write.err
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/${PRE}pmc_sqa -o a -- python3 $P > /dev/null 2> $OUT/${PRE}pmc_sqa.err
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/${PRE}pmc_sqb -o b -- python3 $P > /dev/null 2> $OUT/${PRE}pmc_sqb.err
  unset VBT_PIPELINE_DEPTH
  echo "${PRE}counters done"
}
session ""
# what a small-batch forward is made of: kernel duration and gap per launch position, one stream, hipGraph replay
for nb in 1 8; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_b$nb -o t -- python3 tools/b1_trace.py run $nb > $OUT/tr_b$nb.log 2>&1
  python3 tools/b1_trace.py parse $OUT/tr_b$nb > $OUT/b${nb}_trace.txt 2>&1
done
echo "small-batch traces done"
# BASELINE config 4: EfficientDet-Lite2 448x448 through the same contract run (rehearsal knob VBT_BENCH_MODEL), its own pinned plan
export VBT_BENCH_MODEL=$PWD/models/efficientdet_lite2_synth.vbtm
export VBT_PLAN_FILE=$PWD/profiles/plan_lite2
session "lite2_"
ls $OUT
