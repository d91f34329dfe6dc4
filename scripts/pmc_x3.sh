cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CR_PRECISION=fp32x3
R="rocprofv3 --kernel-trace --output-format csv"
$R --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/pmc/x3_sq -o p -- python3 scripts/pmc_conv.py > /dev/null 2>&1 &&
$R --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD -d gpurun_out/pmc/x3_lds -o p -- python3 scripts/pmc_conv.py > /dev/null 2>&1 &&
$R --pmc FETCH_SIZE -d gpurun_out/pmc/x3_fetch -o p -- python3 scripts/pmc_conv.py > /dev/null 2>&1 &&
$R --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d gpurun_out/pmc/x3_tcc -o p -- python3 scripts/pmc_conv.py > /dev/null 2>&1
ls gpurun_out/pmc
