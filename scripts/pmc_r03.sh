# round-3 PMC passes (separate runs per counter group, as MI355X_MICROARCH.md prescribes): HBM traffic + SQ counters of the grouped
# pyramid convolutions (fp32) and HBM traffic of the argmax-only geometry launch -> gpurun_out/pmc3/*, summarised by pmc_summarize_r03.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R="rocprofv3 --kernel-trace --output-format csv"
G="python3 bench.py --workload geometry --argmax-only --steps 5 --warmup 2 --no-cpu-baseline"
$R --pmc FETCH_SIZE -d gpurun_out/pmc3/grp_fetch -o p -- python3 scripts/pmc_conv_group.py > /dev/null 2>&1 &&
$R --pmc WRITE_SIZE -d gpurun_out/pmc3/grp_write -o p -- python3 scripts/pmc_conv_group.py > /dev/null 2>&1 &&
$R --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/pmc3/grp_sq -o p -- python3 scripts/pmc_conv_group.py > /dev/null 2>&1 &&
$R --pmc FETCH_SIZE -d gpurun_out/pmc3/geoa_fetch -o p -- $G > /dev/null 2>&1 &&
$R --pmc WRITE_SIZE -d gpurun_out/pmc3/geoa_write -o p -- $G > /dev/null 2>&1 &&
python3 scripts/pmc_summarize_r03.py
