# round 4, first look at the one-sweep coder: parity, then timings of both coders on both meshes
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4a
mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_onesweep.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for m in grid walk; do for sw in 1 2; do
  echo "== $m sweeps=$sw"
  TRICO_FPC32_SWEEPS=$sw timeout -k 10 120 python tools/perf_fpc32.py $m > $O/perf_${m}_$sw.log 2>&1 || { tail -20 $O/perf_${m}_$sw.log; exit 1; }
  grep "kernel span" $O/perf_${m}_$sw.log
done; done
