# sweep the number of segments (waves) of the float encoder; usage: gpurun -- bash tools/run_waves.sh "3072 5376" grid
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for kind in ${2:-grid}; do
for w in $1; do
  echo "== $kind waves $w"
  TRICO_FPC32_WAVES=$w timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pw_${kind}_$w -- python $R/tools/perf_fpc32.py $kind > $R/gpurun_out/pw_${kind}_$w.log 2>&1 || exit 1
  grep "kernel span" $R/gpurun_out/pw_${kind}_$w.log; python $R/tools/prof_summary.py $R/gpurun_out/pw_${kind}_$w | head -4
done
done
