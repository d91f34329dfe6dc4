#!/bin/bash
# rocprofv3 kernel durations of the four project+score variants -> gpurun_out/prof_geo_<variant>.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "fast all" "fast none" "exact all" "exact none"; do
  n=$(echo $v | tr ' ' '_')
  rm -rf gpurun_out/prof_geo/$n
  rocprofv3 --kernel-trace -d gpurun_out/prof_geo/$n -o p -- python3 scripts/geo_one.py $v > gpurun_out/prof_geo_$n.log 2>&1 || { echo "rocprof failed for $v"; exit 1; }
  DB=$(ls gpurun_out/prof_geo/$n/*.db gpurun_out/prof_geo/$n/*/*.db 2>/dev/null | head -1)
  [ -n "$DB" ] || { echo "no db for $v"; exit 1; }
  python scripts/rocpd_stats.py $DB gpurun_out/prof_geo_$n.csv > /dev/null 2>&1
  echo "== $v"; grep "k_project_score" gpurun_out/prof_geo_$n.csv < /dev/null
done
rm -rf gpurun_out/prof_geo
