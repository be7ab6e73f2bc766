# per-kernel durations of the one-sweep coder on both meshes
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4b
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for m in grid walk; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc_$m -- python $R/tools/perf_fpc32.py $m > $O/enc_$m.log 2>&1
  echo "## $m"; grep "kernel span" $O/enc_$m.log; python $R/tools/prof_summary.py $O/enc_$m
done > $O/summary.txt
cat $O/summary.txt
rm -rf $O/enc_grid $O/enc_walk
