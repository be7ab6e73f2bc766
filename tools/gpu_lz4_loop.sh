#!/bin/bash
# LZ4 codec: its parity tests, then kernel times on the u32 / u64 triangles of the benchmark meshes
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/lz4_loop
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_lz4_chunked.py tests/test_gpu_parity.py tests/test_gpu_lz4_geometry.py tests/test_gpu_lz4_api.py tests/test_gpu_batch.py -m gpu -x -q --timeout=600 -k "lz4 or u64 or golden or int_edges or mesh or mutated or corrupt or block_api" > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
for m in grid grid64 walk; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$m -- python $R/tools/perf_lz4.py $m > $O/perf_$m.log 2>&1 || { tail -30 $O/perf_$m.log; exit 1; }
  echo "## $m"; grep "encode iter 2\|decode iter 1" $O/perf_$m.log
  python $R/tools/prof_summary.py $O/trace_$m | grep "k_lz4\|k_pd_" | head -14
  rm -rf $O/trace_$m
done
