# class lengths of the sweep from a table (TRICO_FPC32_STAGGER_W, test-hooks builds): bash tools/gpu_weights.sh "w0,w1,..." ...
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/weights
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for w in "$@"; do
  i=$((i+1))
  [ "$w" = "-" ] || export TRICO_FPC32_STAGGER_W=$w
  for m in grid walk; do
    export TRICO_AMD_LIB=$R/tests/_build/libtrico_testhooks.so
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc -- python $R/tools/perf_fpc32.py $m > $O/w${i}_$m.log 2>&1
    echo "## $w $m"; grep "kernel span" $O/w${i}_$m.log; python $R/tools/prof_summary.py $O/enc | grep "sweep<"; rm -rf $O/enc
    export TRICO_AMD_LIB=$R/gpurun_variants/libtrico_diag6.so
    TRICO_SWEEP_DIAG_FILE=$O/diag.bin timeout -k 10 200 python $R/tools/perf_fpc32.py $m > $O/d${i}_$m.log 2>&1 && python $R/tools/diag_analyze.py $O/diag.bin | grep "by arrival\|last end\|component 2\|units with\|latest units"
  done
done
