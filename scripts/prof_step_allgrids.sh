# kernel trace of 20 timed fp32 train steps; per-(kernel, grid) statistics of EVERY kernel -> gpurun_out/prof_grids_all.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CR_PRECISION=fp32 CR_BENCH_BF16=0
rm -rf gpurun_out/prof_g
rocprofv3 --kernel-trace -d gpurun_out/prof_g -o p -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --lean --train-only > gpurun_out/prof_g.log 2>&1
DB=$(ls gpurun_out/prof_g/*.db gpurun_out/prof_g/*/*.db 2>/dev/null | head -1)
python scripts/rocpd_stats.py $DB gpurun_out/prof_grids_all.csv --last-steps 20 --skip-last 3 --by-grid "" >> gpurun_out/prof_g.log 2>&1
rm -rf gpurun_out/prof_g
