# SQ counters of k_project_score (three separate --pmc passes) -> gpurun_out/pmc_geo.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R="rocprofv3 --kernel-trace --output-format csv"
B="python3 bench.py --workload geometry --steps 5 --warmup 2 --no-cpu-baseline"
$R --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES -d gpurun_out/pmc_geo/a -o p -- $B > /dev/null 2>&1 &&
$R --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d gpurun_out/pmc_geo/b -o p -- $B > /dev/null 2>&1 &&
$R --pmc GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC -d gpurun_out/pmc_geo/c -o p -- $B > /dev/null 2>&1
python3 scripts/pmc_parse.py gpurun_out/pmc_geo/a gpurun_out/pmc_geo/b gpurun_out/pmc_geo/c --match k_project_score > gpurun_out/pmc_geo.json
