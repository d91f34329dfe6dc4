cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R="rocprofv3 --kernel-trace --output-format csv"
$R --pmc FETCH_SIZE -d gpurun_out/pmc/f32_fetch -o p -- python3 scripts/pmc_conv.py > /dev/null 2>&1 &&
$R --pmc WRITE_SIZE -d gpurun_out/pmc/f32_write -o p -- python3 scripts/pmc_conv.py > /dev/null 2>&1 &&
$R --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/pmc/f32_sq -o p -- python3 scripts/pmc_conv.py > /dev/null 2>&1 &&
export CR_PRECISION=bf16 &&
$R --pmc FETCH_SIZE -d gpurun_out/pmc/bf16_fetch -o p -- python3 scripts/pmc_conv.py > /dev/null 2>&1 &&
$R --pmc WRITE_SIZE -d gpurun_out/pmc/bf16_write -o p -- python3 scripts/pmc_conv.py > /dev/null 2>&1 &&
$R --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/pmc/bf16_sq -o p -- python3 scripts/pmc_conv.py > /dev/null 2>&1 &&
export CR_PRECISION=fp32x3 &&
$R --pmc FETCH_SIZE -d gpurun_out/pmc/x3_fetch -o p -- python3 scripts/pmc_conv.py > /dev/null 2>&1 &&
$R --pmc WRITE_SIZE -d gpurun_out/pmc/x3_write -o p -- python3 scripts/pmc_conv.py > /dev/null 2>&1 &&
$R --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/pmc/x3_sq -o p -- python3 scripts/pmc_conv.py > /dev/null 2>&1 &&
unset CR_PRECISION &&
$R --pmc FETCH_SIZE -d gpurun_out/pmc/geo_fetch -o p -- python3 bench.py --workload geometry --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 &&
$R --pmc WRITE_SIZE -d gpurun_out/pmc/geo_write -o p -- python3 bench.py --workload geometry --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc/geo_stats -o p -- python3 bench.py --workload geometry --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/pmc/geo_bench.json 2>/dev/null
ls gpurun_out/pmc
