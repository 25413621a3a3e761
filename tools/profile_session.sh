#!/bin/bash
# One measurement session on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_session.sh r02
# writes raw outputs under gpurun_out/<tag>/ ; tools/make_profile_summary.py <tag> turns them into profiles/<tag>_*.
# Counter passes are separate rocprofv3 runs with --kernel-trace only (no other trace domain), depth 1, pinned plan.
set -u
TAG=${1:-r02}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
export VBT_PLAN_FILE=$PWD/profiles/plan_lite0
OUT=gpurun_out/$TAG
mkdir -p $OUT
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 bench.py --steps 20 --warmup 5 --cpu-frames 0 > $OUT/bench_k20.json 2> $OUT/bench_k20.err
B="bench.py --steps 50 --warmup 5 --cpu-frames 0 --no-extras --no-roofline --settle-steps 0"
VBT_PIPELINE_DEPTH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_d1 -o d1 -- python3 $B > /dev/null 2> $OUT/trace_d1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_d3 -o d3 -- python3 $B > /dev/null 2> $OUT/trace_d3.err
P="bench.py --steps 6 --warmup 2 --cpu-frames 0 --no-extras --no-roofline --settle-steps 0"
export VBT_PIPELINE_DEPTH=1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 $P > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 $P > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_sqa -o a -- python3 $P > /dev/null 2> $OUT/pmc_sqa.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_sqb -o b -- python3 $P > /dev/null 2> $OUT/pmc_sqb.err
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_grbm -o g -- python3 $P > /dev/null 2> $OUT/pmc_grbm.err
ls $OUT
