# kernel trace of 20 timed train steps in the precision mode given as $1 (fp32 | fp32x3 | bf16) -> gpurun_out/prof_$1.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CR_PRECISION=$1 CR_BENCH_BF16=0
rm -rf gpurun_out/prof_$1
rocprofv3 --kernel-trace -d gpurun_out/prof_$1 -o p -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --lean --train-only > gpurun_out/prof_$1.log 2>&1
DB=$(ls gpurun_out/prof_$1/*.db gpurun_out/prof_$1/*/*.db 2>/dev/null | head -1)
python scripts/rocpd_stats.py $DB gpurun_out/prof_$1.csv --last-steps 20 --skip-last 3 >> gpurun_out/prof_$1.log 2>&1
rm -rf gpurun_out/prof_$1
