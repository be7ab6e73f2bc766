# LZ4 chunk geometry sweep: same bytes (sizes, decode ok) whatever the geometry; time and stitch statistics per setting
cd $GRAFT_REPO_ROOT
MESH=${1:-grid}
shift || true
CFGS=${@:-"1048576,393216 131072,131072 262144,70000 196608,98304 524288,131072"}
for cfg in $CFGS; do
  c=${cfg%,*}; w=${cfg#*,}
  echo "== chunk $c warm $w"
  TRICO_LZ4_CHUNK=$c TRICO_LZ4_WARM=$w timeout -k 10 100 python tools/perf_lz4.py $MESH 2>&1 | grep "encode iter 2\|decode iter 1"
done
