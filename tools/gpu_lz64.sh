#!/bin/bash
# LZ4 codec on the u64 triangles of config 3 (8 byte planes) against the u32 ones: kernel times and chunk statistics
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/lz64
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in grid grid64; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$m -- python $R/tools/perf_lz4.py $m > $O/perf_$m.log 2>&1 || { tail -30 $O/perf_$m.log; exit 1; }
  echo "## $m"; grep "encode iter 2\|decode iter 1" $O/perf_$m.log
  python $R/tools/prof_summary.py $O/trace_$m | grep "k_lz4\|k_pd\|k_planes" | head -12
  rm -rf $O/trace_$m
done
TRICO_LZ4_DEBUG=1 timeout -k 10 300 python $R/tools/perf_lz4.py grid64 2>&1 | grep -A1 "^plane" | tail -20
