set -e
cd $GRAFT_REPO_ROOT
for kind in grid walk; do
  echo "== variant 4 $kind"
  TRICO_FPC32_DEC=4 timeout -k 10 200 python tools/perf_fpc32_decode.py $kind 4000 2500 2>&1 | grep "comp\|prof"
done
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_d.log 2>&1 || { tail -30 gpurun_out/pytest_d.log; exit 1; }
tail -3 gpurun_out/pytest_d.log
