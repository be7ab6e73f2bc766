set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/dbl_loop
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lowlevel.py tests/test_gpu_batch.py tests/test_gpu_selfcheck.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 300 python tools/perf_config3.py > $O/c3.log 2>&1 || { tail -30 $O/c3.log; exit 1; }
tail -1 $O/c3.log
