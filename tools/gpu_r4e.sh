# SQ counters of the one-sweep coder (asm step) on both meshes + timings, lag / prio variants
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4e
mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_onesweep.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
cd /tmp; export TMPDIR=/tmp
for m in grid walk; do
  for lp in "0 1" "1 1" "0 0"; do
    set -- $lp
    TRICO_FPC32_LAG=$1 TRICO_FPC32_PRIO=$2 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc -- python $R/tools/perf_fpc32.py $m > $O/enc.log 2>&1
    echo "## $m lag=$1 prio=$2"; grep "kernel span" $O/enc.log; python $R/tools/prof_summary.py $O/enc | grep -v "selftest\|rocclr\|pscan\|offsets"
    rm -rf $O/enc
  done
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $O/pmc1 -- python $R/tools/perf_fpc32.py $m > $O/pmc1.log 2>&1
  python $R/tools/pmc_summary.py $O/pmc1 | grep -A9 "k_fpc32_sweep"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc2 -- python $R/tools/perf_fpc32.py $m > $O/pmc2.log 2>&1
  python $R/tools/pmc_summary.py $O/pmc2
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc3 -- python $R/tools/perf_fpc32.py $m > $O/pmc3.log 2>&1
  python $R/tools/pmc_summary.py $O/pmc3
  rm -rf $O/pmc1 $O/pmc2 $O/pmc3
done > $O/summary.txt 2>&1
cat $O/summary.txt
