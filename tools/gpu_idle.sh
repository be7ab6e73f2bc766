# the float encoder with an idle device in front of every encode: bash tools/gpu_idle.sh
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for ms in 0 1 20 300; do
  echo "## idle $ms ms"; PERF_IDLE_MS=$ms timeout -k 10 200 python $R/tools/perf_fpc32.py grid | grep "kernel span"
done
