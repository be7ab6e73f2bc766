# LZ4 chunk geometry sweep: same bytes (sizes, decode ok) whatever the geometry; time and stitch statistics per setting
cd $GRAFT_REPO_ROOT
MESH=${1:-grid}
for cfg in "1048576 393216" "131072 131072" "262144 70000" "196608 98304" "524288 131072"; do
  set -- $cfg
  echo "== chunk $1 warm $2"
  TRICO_LZ4_CHUNK=$1 TRICO_LZ4_WARM=$2 timeout -k 10 100 python tools/perf_lz4.py $MESH 2>&1 | grep "encode iter 2\|decode iter 1"
done
