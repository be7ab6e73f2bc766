set -e
R=$GRAFT_REPO_ROOT
cd $R
for e in "" no_t2_store no_stores; do
  echo "## experiment: ${e:-none}"
  if [ -n "$e" ]; then export TRICO_AMD_LIB=$R/tools/_exp/libtrico_$e.so; else unset TRICO_AMD_LIB; fi
  timeout -k 10 200 python tools/perf_config3_nocheck.py 2>&1 | tail -2
done
echo "## float chain, store behind the load"
export TRICO_AMD_LIB=$R/tools/_exp/libtrico_ch5_store_after_load.so
timeout -k 10 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --concurrent "" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['decode_model']['ns_per_value'], j['config']['parity'])"
