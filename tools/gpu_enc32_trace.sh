#!/bin/bash
# Float-vertex encoder: its parity tests (unless "quick"), then start / end of every kernel of the last encodes
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/enc32
mkdir -p $O
cd $R
if [ "$1" != quick ]; then
  timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_onesweep.py tests/test_gpu_batch.py tests/test_gpu_lowlevel.py -m gpu -x -q --timeout=600 > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
  tail -2 $O/pytest.log
fi
cd /tmp && export TMPDIR=/tmp
for m in grid walk; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$m -- python $R/tools/perf_fpc32.py $m > $O/perf_$m.log 2>&1 || { tail -30 $O/perf_$m.log; exit 1; }
  echo "## $m"; grep "kernel span" $O/perf_$m.log
  python $R/tools/prof_summary.py $O/trace_$m | grep k_fpc32 | tee $O/kernels_$m.txt
  python $R/tools/trace_summary.py $O/trace_$m k_fpc32 | tail -14 > $O/dispatches_$m.txt
  rm -rf $O/trace_$m
done
