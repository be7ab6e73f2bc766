# workgroups of the sweep per compute unit (TRICO_FPC32_SPARE, test-hooks library): sweep and gather times.  bash tools/gpu_spare.sh 1 2 ...
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
export TRICO_AMD_LIB=$R/tests/_build/libtrico_testhooks.so
for sp in "$@"; do
  export TRICO_FPC32_SPARE=$sp
  for m in grid walk; do
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/enc_sp -- python $R/tools/perf_fpc32.py $m > /tmp/sp.log 2>&1
    echo "## spare $sp $m: $(grep 'kernel span' /tmp/sp.log | cut -c1-30)"; python $R/tools/prof_summary.py /tmp/enc_sp | grep 'sweep<\|gather\|scanfix'; rm -rf /tmp/enc_sp
  done
done
