R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
export TRICO_AMD_LIB=$R/tests/_build/libtrico_testhooks.so
for sp in 1 0 2; do for b in 300 500; do
  export TRICO_FPC32_SPARE=$sp TRICO_FPC32_STAGGER=$b
  for m in grid walk; do
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/enc_sp -- python $R/tools/perf_fpc32.py $m > /tmp/sp.log 2>&1
    echo "## spare $sp beta $b $m: $(grep 'kernel span' /tmp/sp.log | cut -c1-30) $(python $R/tools/prof_summary.py /tmp/enc_sp | grep 'sweep<' | cut -c28-60)"; rm -rf /tmp/enc_sp
  done
done; done
