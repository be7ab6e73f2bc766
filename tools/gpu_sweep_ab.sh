# the float encoder after a change of the sweep: parity with the compiled step and with the hand-written loop, then timings
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/sweep_ab
mkdir -p $O
cd $R
echo "== compiled step (TRICO_FPC32_ASM=0)"
TRICO_AMD_LIB=$R/tests/_build/libtrico_testhooks.so TRICO_FPC32_ASM=0 timeout -k 10 400 python -m pytest tests/test_gpu_onesweep.py tests/test_gpu_parity.py -m gpu -x -q -k "(fp or sweep or mesh or golden or config1 or guard or sentinel or bunny) and not product_library" > $O/pytest_c.log 2>&1 || { tail -40 $O/pytest_c.log; exit 1; }
tail -2 $O/pytest_c.log
echo "== hand-written loop"
timeout -k 10 400 python -m pytest tests/test_gpu_onesweep.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_a.log 2>&1 || { tail -60 $O/pytest_a.log; exit 1; }
tail -2 $O/pytest_a.log
cd /tmp; export TMPDIR=/tmp
for m in grid walk; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc -- python $R/tools/perf_fpc32.py $m > $O/enc_$m.log 2>&1
  echo "## $m"; grep "kernel span" $O/enc_$m.log; python $R/tools/prof_summary.py $O/enc | grep -v "selftest\|rocclr"
  rm -rf $O/enc
done
