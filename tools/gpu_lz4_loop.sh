# LZ4 codec: parity tests, then encode / decode times on both meshes
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/lz4_loop
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_lz4_geometry.py tests/test_gpu_lz4_chunked.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for m in grid walk; do
  echo "== $m"
  timeout -k 10 200 python tools/perf_lz4.py $m 2>&1 | grep -E "encode iter 2|decode iter 1"
done
