# parity + timings + HBM traffic for lag variants
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/enc_loop
mkdir -p $O
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_onesweep.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
cd /tmp; export TMPDIR=/tmp
for m in grid walk; do
  for lag in 8; do
    export TRICO_FPC32_LAG=$lag
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc -- python $R/tools/perf_fpc32.py $m > $O/enc.log 2>&1
    echo "## $m lag=$lag"; grep "kernel span" $O/enc.log; python $R/tools/prof_summary.py $O/enc | grep -v "selftest\|rocclr\|pscan\|offsets"
    rm -rf $O/enc
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc2 -- python $R/tools/perf_fpc32.py $m > $O/pmc2.log 2>&1
    python $R/tools/pmc_summary.py $O/pmc2 | grep -A1 "k_fpc32_sweep\|k_fpc32_gather"
    rm -rf $O/pmc2
  done
done > $O/summary.txt 2>&1
cat $O/summary.txt
