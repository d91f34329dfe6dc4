cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_dom
CR_BENCH_BF16=0 rocprofv3 --kernel-trace -d gpurun_out/prof_dom -o p -- python3 bench.py --no-cpu-baseline > gpurun_out/dom_bench.json 2> gpurun_out/dom_bench.err
DB=$(ls gpurun_out/prof_dom/*.db gpurun_out/prof_dom/*/*.db 2>/dev/null | head -1)
python scripts/dominant_instances.py $DB gpurun_out/dom_bench.json gpurun_out/dominant.json
rm -rf gpurun_out/prof_dom
