# per-kernel times of the float encoder for the measurement builds in gpurun_variants/ (tools/build_variant.sh): bash tools/gpu_variants.sh A B ...
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/variants
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  export TRICO_AMD_LIB=$R/gpurun_variants/libtrico_$v.so
  for m in grid walk; do
    case $v in
      *diag) timeout -k 10 120 python $R/tools/perf_fpc32.py $m > $O/${v}_$m.log 2>&1; echo "## $v $m"; grep "kernel span" $O/${v}_$m.log; grep "sweep diag" $O/${v}_$m.log | tail -3 ;;
      *) timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc -- python $R/tools/perf_fpc32.py $m > $O/${v}_$m.log 2>&1
         echo "## $v $m"; grep "kernel span" $O/${v}_$m.log; python $R/tools/prof_summary.py $O/enc | grep "sweep<\|gather\|scanfix\|offsets"; rm -rf $O/enc ;;
    esac
  done
done
