# one-sweep coder with the hand-written step: parity, then kernel breakdown asm vs compiled on both meshes
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4d
mkdir -p $O
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_onesweep.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
cd /tmp; export TMPDIR=/tmp
for m in grid walk; do for a in 1 0; do
  TRICO_FPC32_ASM=$a timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc -- python $R/tools/perf_fpc32.py $m > $O/enc.log 2>&1
  echo "## $m asm=$a"; grep "kernel span" $O/enc.log; python $R/tools/prof_summary.py $O/enc | grep -v "selftest\|rocclr"
  rm -rf $O/enc
done; done > $O/summary.txt
cat $O/summary.txt
