# kernel trace of the weak train step (bench.py --workload weak): per-kernel statistics of the timed steps -> gpurun_out/prof_weak.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_weak
rocprofv3 --kernel-trace -d gpurun_out/prof_weak -o p -- python3 bench.py --workload weak --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_weak.log 2>&1
DB=$(ls gpurun_out/prof_weak/*.db gpurun_out/prof_weak/*/*.db 2>/dev/null | head -1)
python scripts/rocpd_stats.py $DB gpurun_out/prof_weak.csv --last-steps 20 >> gpurun_out/prof_weak.log 2>&1
rm -rf gpurun_out/prof_weak
