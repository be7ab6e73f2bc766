cd $GRAFT_REPO_ROOT
for cfg in "1048576 393216" "1048576 786432" "1048576 1048576" "2097152 1048576" "2097152 2097152" "4194304 2097152"; do
  set -- $cfg
  echo "== chunk $1 warm $2"
  TRICO_LZ4_CHUNK=$1 TRICO_LZ4_WARM=$2 timeout -k 10 100 python tools/perf_lz4.py grid 2>&1 | grep "encode iter 2"
done
