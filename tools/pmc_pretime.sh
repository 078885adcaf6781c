cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_pt
mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/a -o a -- python3 $R/tools/pretime_bench.py 32 3 12 100 32 1 > /dev/null 2>&1
python3 $R/tools/pmc_generic.py $O/a/a_counter_collection.csv pretime_kernel
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/b -o b -- python3 $R/tools/pretime_bench.py 32 3 12 100 32 1 > /dev/null 2>&1
python3 $R/tools/pmc_generic.py $O/b/b_counter_collection.csv pretime_kernel
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_TRANS SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/c -o c -- python3 $R/tools/pretime_bench.py 32 3 12 100 32 1 > /dev/null 2>&1
python3 $R/tools/pmc_generic.py $O/c/c_counter_collection.csv pretime_kernel
rm -rf $O/a $O/b $O/c
