# float-vertex encoder: its parity tests, then per-kernel times on both meshes (product library).  bash tools/gpu_enc32.sh [quick]
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/enc32
mkdir -p $O
cd $R
if [ "$1" != quick ]; then
  timeout -k 10 600 python -m pytest tests/test_gpu_onesweep.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
  tail -2 $O/pytest.log
fi
cd /tmp; export TMPDIR=/tmp
for m in grid walk; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc -- python $R/tools/perf_fpc32.py $m > $O/$m.log 2>&1
  echo "## $m"; grep "kernel span" $O/$m.log; python $R/tools/prof_summary.py $O/enc | grep "k_fpc32"; rm -rf $O/enc
done 2>&1 | tee $O/summary.txt
