cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp VBT_PLAN_FILE=$PWD/profiles/plan_lite0 VBT_PIPELINE_DEPTH=1
OUT=gpurun_out/sq2
mkdir -p $OUT
P="bench.py --steps 6 --warmup 2 --cpu-frames 0 --no-extras --no-roofline"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_sqa -o a -- python3 $P > /dev/null 2> $OUT/pmc_sqa.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_sqb -o b -- python3 $P > /dev/null 2> $OUT/pmc_sqb.err
ls $OUT
