# round-2 measurement batch A: instruction-cost microbenchmark, GPU tests, float decoder variants
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 $R/tools/ubench/lat.hip -o /tmp/lat 2>/dev/null
timeout -k 10 60 /tmp/lat > $O/lat.log 2>&1
cat $O/lat.log
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_a.log 2>&1 || { tail -30 $O/pytest_a.log; exit 1; }
tail -3 $O/pytest_a.log
for v in 2 3 1; do
  for kind in grid walk; do
    echo "== variant $v $kind"
    TRICO_FPC32_DEC=$v timeout -k 10 200 python tools/perf_fpc32_decode.py $kind 2>&1 | grep comp
  done
done
