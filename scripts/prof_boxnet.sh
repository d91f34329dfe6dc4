# kernel trace of the BoxNet pipeline (bench.py --workload boxnet) -> gpurun_out/prof_boxnet.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_boxnet
rocprofv3 --kernel-trace -d gpurun_out/prof_boxnet -o p -- python3 bench.py --workload boxnet --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_boxnet.log 2>&1
DB=$(ls gpurun_out/prof_boxnet/*.db gpurun_out/prof_boxnet/*/*.db 2>/dev/null | head -1)
python scripts/rocpd_stats.py $DB gpurun_out/prof_boxnet.csv >> gpurun_out/prof_boxnet.log 2>&1
rm -rf gpurun_out/prof_boxnet
