#!/bin/bash
# config 3: where the encode's time goes (kernel trace of tools/perf_config3.py; the decode dominates the run, the summary lists encoders)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c3enc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/tools/perf_config3.py > $O/perf.log 2>&1 || { tail -30 $O/perf.log; exit 1; }
tail -1 $O/perf.log
python $R/tools/prof_summary.py $O/trace | grep -v "decode\|k_pd_\|chain" | head -40 | tee $O/kernels.txt
rm -rf $O/trace
