# wave-priority experiment of the float encoder's code sweep; usage: gpurun -- bash tools/run_prio.sh "0 3 8" "grid walk"
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for kind in ${2:-grid}; do
for m in $1; do
  echo "== $kind prio $m"
  TRICO_FPC32_PRIO=$m timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pp_${kind}_$m -- python $R/tools/perf_fpc32.py $kind > $R/gpurun_out/pp_${kind}_$m.log 2>&1 || exit 1
  grep "kernel span" $R/gpurun_out/pp_${kind}_$m.log; python $R/tools/prof_summary.py $R/gpurun_out/pp_${kind}_$m | head -3
  rm -rf $R/gpurun_out/pp_${kind}_$m
done
done
