# SQ counters of the kernels matching $1 for the command in $2... (three passes); prints per-kernel medians
# usage: bash scripts/pmc_kernel.sh k_roi_bwd_tiles python3 scripts/roi_tiles_bench.py
M=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R="rocprofv3 --kernel-trace --output-format csv"
rm -rf gpurun_out/pmc_k
$R --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES -d gpurun_out/pmc_k/a -o p -- "$@" > /dev/null 2>&1
$R --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d gpurun_out/pmc_k/b -o p -- "$@" > /dev/null 2>&1
$R --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INST_CYCLES_SALU -d gpurun_out/pmc_k/c -o p -- "$@" > /dev/null 2>&1
python3 scripts/pmc_parse.py gpurun_out/pmc_k/a gpurun_out/pmc_k/b gpurun_out/pmc_k/c --match $M
