# one-sweep vs two-sweep float encoder; usage: gpurun -- bash tools/run_sweeps.sh "2 1" "grid walk"
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for kind in ${2:-grid}; do
for m in $1; do
  echo "== $kind sweeps $m"
  TRICO_FPC32_SWEEPS=$m timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ps_${kind}_$m -- python $R/tools/perf_fpc32.py $kind > $R/gpurun_out/ps_${kind}_$m.log 2>&1 || exit 1
  grep "kernel span" $R/gpurun_out/ps_${kind}_$m.log; python $R/tools/prof_summary.py $R/gpurun_out/ps_${kind}_$m > $R/gpurun_out/ps_${kind}_$m.txt; head -9 $R/gpurun_out/ps_${kind}_$m.txt
  rm -rf $R/gpurun_out/ps_${kind}_$m
done
done
