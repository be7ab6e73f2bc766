# the float encoder with fewer, longer segments: workgroups per compute unit = 10 - TRICO_FPC32_SPARE (test-hooks build): bash tools/gpu_spare.sh 1 3 5 ...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/spare; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export TRICO_AMD_LIB=$R/tests/_build/libtrico_testhooks.so
for sp in "$@"; do
  for m in grid walk; do
    echo "## spare $sp $m"
    TRICO_FPC32_SPARE=$sp timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc -- python $R/tools/perf_fpc32.py $m > $O/enc.log 2>&1
    grep "kernel span" $O/enc.log; python $R/tools/prof_summary.py $O/enc | grep "sweep\|gather\|fixup"; rm -rf $O/enc
  done
done
