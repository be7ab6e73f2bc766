#!/bin/bash
# Double encoder (k_fpc64_sort.hip): its parity tests, then kernel times of the two vec3 double streams of config 3.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/enc64
mkdir -p $O
cd $R
[ "$1" = quick ] || timeout -k 10 900 python -m pytest tests/test_gpu_fpc64_encoder.py tests/test_gpu_parity.py -m gpu -x -q --timeout=300 -k "fp64 or fpc64 or encoder or multi or bunny or golden" > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
[ "$1" = quick ] || tail -2 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/tools/perf_fpc64.py > $O/perf.log 2>&1 || { tail -30 $O/perf.log; exit 1; }
grep "kernel span\|iter 3" $O/perf.log
python $R/tools/prof_summary.py $O/trace | tee $O/kernels.txt
python $R/tools/trace_summary.py $O/trace k64_ | tail -24 > $O/dispatches.txt
rm -rf $O/trace
