# tile-variant experiment of the float encoder; usage: gpurun -- bash tools/run_tile.sh "0 1 3" "grid walk"
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for kind in ${2:-grid}; do
for m in $1; do
  echo "== $kind tile $m"
  TRICO_FPC32_TILE=$m TRICO_FPC32_PRIO=8 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pt_${kind}_$m -- python $R/tools/perf_fpc32.py $kind > $R/gpurun_out/pt_${kind}_$m.log 2>&1 || exit 1
  grep "kernel span" $R/gpurun_out/pt_${kind}_$m.log; python $R/tools/prof_summary.py $R/gpurun_out/pt_${kind}_$m | head -3
  rm -rf $R/gpurun_out/pt_${kind}_$m
done
done
