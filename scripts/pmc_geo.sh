# SQ counters of k_project_score (two passes), for the shipped kernel and the timing-experiment skeleton
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R="rocprofv3 --kernel-trace --output-format csv"
for e in ${GEO_EXPS:-0 31}; do
export CR_GEO_EXP=$e
$R --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES -d gpurun_out/pmc_geo/e${e}_a -o p -- python3 bench.py --workload geometry --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 &&
$R --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d gpurun_out/pmc_geo/e${e}_b -o p -- python3 bench.py --workload geometry --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 &&
$R --pmc GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY -d gpurun_out/pmc_geo/e${e}_c -o p -- python3 bench.py --workload geometry --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python3 scripts/pmc_parse.py gpurun_out/pmc_geo/e${e}_a gpurun_out/pmc_geo/e${e}_b gpurun_out/pmc_geo/e${e}_c --match k_project_score > gpurun_out/pmc_geo_e${e}.json
done
unset CR_GEO_EXP
