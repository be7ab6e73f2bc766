set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4k
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for t in 256 128 512; do
  if [ $t = 256 ]; then unset TRICO_AMD_LIB; else export TRICO_AMD_LIB=$R/tools/_exp/libtrico_gt$t.so; fi
  for m in grid walk; do
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc -- python $R/tools/perf_fpc32.py $m > $O/enc.log 2>&1
    echo "## threads $t $m"; grep "kernel span" $O/enc.log; python $R/tools/prof_summary.py $O/enc | grep "gather"
    rm -rf $O/enc
  done
done
