# one-sweep coder: variants of lag / prio, kernel breakdown + SQ counters on the grid mesh
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c
mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_onesweep.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
cd /tmp; export TMPDIR=/tmp
for m in grid walk; do for lag in 0 1; do for prio in 0 1; do
  TRICO_FPC32_LAG=$lag TRICO_FPC32_PRIO=$prio timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc -- python $R/tools/perf_fpc32.py $m > $O/enc.log 2>&1
  echo "## $m lag=$lag prio=$prio"; grep "kernel span" $O/enc.log; python $R/tools/prof_summary.py $O/enc | grep -E "sweep|gather"
  rm -rf $O/enc
done; done; done > $O/summary.txt
cat $O/summary.txt
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/pmc_sq1 -- python $R/tools/perf_fpc32.py grid > $O/pmc_sq1.log 2>&1
python $R/tools/pmc_summary.py $O/pmc_sq1 > $O/sq_grid.txt
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/pmc_sq2 -- python $R/tools/perf_fpc32.py walk > $O/pmc_sq2.log 2>&1
python $R/tools/pmc_summary.py $O/pmc_sq2 > $O/sq_walk.txt
rm -rf $O/pmc_sq1 $O/pmc_sq2
grep -A7 "k_fpc32_sweep\|k_fpc32_gather" $O/sq_grid.txt
