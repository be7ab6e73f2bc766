set -e
R=$GRAFT_REPO_ROOT
cd $R
for g in "default" "524288 98304" "655360 98304" "786432 98304" "655360 131072" "1048576 98304"; do
  echo "== geometry $g"
  if [ "$g" = "default" ]; then unset TRICO_LZ4_CHUNK TRICO_LZ4_WARM; else set -- $g; export TRICO_LZ4_CHUNK=$1 TRICO_LZ4_WARM=$2; fi
  timeout -k 10 200 python tools/perf_lz4.py walk 2>&1 | grep -E "encode iter 2|decode iter 1"
done
