cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export VBT_PLAN_FILE=$PWD/profiles/plan_lite0
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
grep -o "SQ_[A-Z_0-9]*" gpurun_out/counters_list.txt | sort -u | tr '\n' ' ' | head -c 6000
echo
B="bench.py --steps 6 --warmup 2 --cpu-frames 0 --no-extras --no-roofline"
VBT_PIPELINE_DEPTH=1 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d gpurun_out/pmc_a -o a --output-format csv -- python3 $B > /dev/null 2> gpurun_out/pmc_a.err
VBT_PIPELINE_DEPTH=1 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA -d gpurun_out/pmc_b -o b --output-format csv -- python3 $B > /dev/null 2> gpurun_out/pmc_b.err
VBT_PIPELINE_DEPTH=1 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVE_DEP_WAIT SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_IFETCH -d gpurun_out/pmc_c -o c --output-format csv -- python3 $B > /dev/null 2> gpurun_out/pmc_c.err
ls gpurun_out/pmc_a gpurun_out/pmc_b gpurun_out/pmc_c; tail -3 gpurun_out/pmc_c.err
