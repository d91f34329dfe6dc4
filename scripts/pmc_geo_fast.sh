# HBM traffic of the fast project+score kernel (full outputs and argmax-only): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
# passes (MI355X_MICROARCH.md), summarised into profiles/r03_pmc_geometry_fast{,_argmax}_traffic.json; kernel durations of the
# four variants without counters -> profiles/r03_geometry_kernel_durations.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R="rocprofv3 --kernel-trace --output-format csv"
for v in all none; do
  $R --pmc FETCH_SIZE -d gpurun_out/pmcg/${v}_fetch -o p -- python3 scripts/geo_one.py fast $v > /dev/null 2>&1 || exit 1
  $R --pmc WRITE_SIZE -d gpurun_out/pmcg/${v}_write -o p -- python3 scripts/geo_one.py fast $v > /dev/null 2>&1 || exit 1
  echo "pmc $v done"
done
for v in "fast all" "fast none" "exact all" "exact none"; do
  n=$(echo $v | tr ' ' '_')
  $R -d gpurun_out/pmcg/dur_$n -o p -- python3 scripts/geo_one.py $v > /dev/null 2>&1 || exit 1
  echo "trace $v done"
done
python3 scripts/pmc_geo_fast_summarize.py
