# kernel trace of the inference workload (bench.py --workload inference) -> gpurun_out/prof_infer.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_infer
rocprofv3 --kernel-trace -d gpurun_out/prof_infer -o p -- python3 bench.py --workload inference --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_infer.log 2>&1
DB=$(ls gpurun_out/prof_infer/*.db gpurun_out/prof_infer/*/*.db 2>/dev/null | head -1)
python scripts/rocpd_stats.py $DB gpurun_out/prof_infer.csv --last-steps 20 --marker k_cube_decode_infer >> gpurun_out/prof_infer.log 2>&1
python scripts/rocpd_stats.py $DB gpurun_out/prof_infer_by_grid.csv --last-steps 20 --marker k_cube_decode_infer --by-grid k_conv >> gpurun_out/prof_infer.log 2>&1
rm -rf gpurun_out/prof_infer
