#!/bin/bash
# LZ4 decompress: duration of every pointer-jumping round (which rounds work, which return at once)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pdjump
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in ${@:-grid walk}; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$m -- python $R/tools/perf_lz4.py $m > $O/perf_$m.log 2>&1
  echo "## $m"; grep "decode iter 1" $O/perf_$m.log
  python $R/tools/trace_summary.py $O/trace_$m k_pd_ | tail -38 | awk '{print $1, $NF-1 " " $(NF-1)}' | tr '\n' ';'
  echo
  rm -rf $O/trace_$m
done
