# PMC passes of the stem patch kernels (forward, backward-data, weight gradient) (scripts/stem_bench.py): HBM traffic and MFMA utilisation
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R="rocprofv3 --kernel-trace --output-format csv"
rm -rf gpurun_out/pmc_stem
$R --pmc FETCH_SIZE -d gpurun_out/pmc_stem/fetch -o p -- python3 scripts/stem_bench.py > /dev/null 2>&1 &&
$R --pmc WRITE_SIZE -d gpurun_out/pmc_stem/write -o p -- python3 scripts/stem_bench.py > /dev/null 2>&1 &&
$R --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/pmc_stem/sq -o p -- python3 scripts/stem_bench.py > /dev/null 2>&1 &&
python3 scripts/pmc_parse.py gpurun_out/pmc_stem/fetch gpurun_out/pmc_stem/write gpurun_out/pmc_stem/sq --match _patch_f32 > gpurun_out/pmc_stem.json
rm -rf gpurun_out/pmc_stem
