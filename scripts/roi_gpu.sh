python -m pytest tests/test_gpu_roi_bwd_tiles.py -x -q > gpurun_out/r3_t8_pytest.log 2>&1; tail -3 gpurun_out/r3_t8_pytest.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/roi_prof -o p -- python3 scripts/roi_tiles_bench.py > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/roi_prof/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "roi" in r["Name"]:
        print(r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
