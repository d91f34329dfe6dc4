# kernel trace of a probe script ($1) -> gpurun_out/prof_probe.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_probe
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_probe -o p --output-format csv -- python3 $1 > gpurun_out/prof_probe.log 2>&1 < /dev/null
cp $(ls gpurun_out/prof_probe/*kernel_stats.csv gpurun_out/prof_probe/*/*kernel_stats.csv 2>/dev/null | head -1) gpurun_out/prof_probe.csv
rm -rf gpurun_out/prof_probe
