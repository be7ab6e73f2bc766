# lag coupling of the code sweep's component waves: time and HBM reads; usage: gpurun -- bash tools/run_lag.sh "0 1 2 4" "grid walk"
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for kind in ${2:-grid}; do for l in $1; do
  echo "== $kind lag $l"
  for rep in 1 2; do TRICO_FPC32_LAG=$l timeout -k 10 120 python $R/tools/perf_fpc32.py $kind | grep "kernel span"; done
  TRICO_FPC32_LAG=$l timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/lag_${kind}_$l -- python $R/tools/perf_fpc32.py $kind > /dev/null 2>&1 || exit 1
  python $R/tools/pmc_summary.py $R/gpurun_out/lag_${kind}_$l | grep -A1 "k_fpc32_code" | tail -1
  rm -rf $R/gpurun_out/lag_${kind}_$l
done; done
