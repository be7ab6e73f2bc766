# float encoder variants; usage: gpurun -- bash tools/run_enc_var.sh grid "TRICO_FPC32_SWEEPS=2" "TRICO_FPC32_WAVES=5376" ...
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
kind=$1; shift
i=0
for v in "$@"; do
  i=$((i+1))
  echo "== $kind $v"
  env $v timeout -k 10 200 python $R/tools/perf_fpc32.py $kind > $R/gpurun_out/pv_${kind}_$i.log 2>&1 || { tail -3 $R/gpurun_out/pv_${kind}_$i.log; exit 1; }
  grep "kernel span" $R/gpurun_out/pv_${kind}_$i.log
done
