mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]+|GRBM_[A-Z_]+|TCC_[A-Z_0-9]+\[?)" | sort -u | tr '\n' ' ' | head -c 6000 > $R/gpurun_out/pmc_list.txt
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/pmc_$name -o p -- python $R/tools/conv_microbench.py up2 --batch 2 --iters 10 > $R/gpurun_out/pmc_$name.log 2>&1; }
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
run b SQ_INSTS_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
ls $R/gpurun_out/pmc_a $R/gpurun_out/pmc_b; tail -2 $R/gpurun_out/pmc_a.log
