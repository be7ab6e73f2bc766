set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4g
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for mesh in grid walk; do
for w in 7680 6144 3072; do
  TRICO_FPC32_WAVES=$w timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc -- python $R/tools/perf_fpc32.py $mesh > $O/enc.log 2>&1 || true
  echo "## $mesh waves=$w"; grep "kernel span" $O/enc.log; python $R/tools/prof_summary.py $O/enc | grep "fpc32"
  rm -rf $O/enc
done
done
