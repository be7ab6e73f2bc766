# the sweep's staggered segment lengths (TRICO_FPC32_STAGGER, test-hooks library): per-kernel times by beta.  bash tools/gpu_stagger.sh 0 150 250 ...
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/stagger
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export TRICO_AMD_LIB=$R/tests/_build/libtrico_testhooks.so
for b in "$@"; do
  export TRICO_FPC32_STAGGER=$b
  for m in grid walk; do
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc -- python $R/tools/perf_fpc32.py $m > $O/b${b}_$m.log 2>&1
    echo "## beta $b $m"; grep "kernel span" $O/b${b}_$m.log; python $R/tools/prof_summary.py $O/enc | grep "sweep<\|gather\|scanfix\|offsets"; rm -rf $O/enc
  done
done 2>&1 | tee $O/summary.txt
