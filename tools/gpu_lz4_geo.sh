# LZ4 compress: encode time of a mesh's index planes by chunk geometry (TRICO_LZ4_CHUNK / _WARM fix one geometry for every plane): bash tools/gpu_lz4_geo.sh MESH "chunk warm" ...
cd /tmp
M=$1; shift
for cfg in "$@"; do
  set -- $cfg
  echo "== $M chunk $1 warm $2: $(TRICO_LZ4_CHUNK=$1 TRICO_LZ4_WARM=$2 timeout 120 python $GRAFT_REPO_ROOT/tools/perf_lz4.py $M 2>&1 | grep 'encode iter 2' | cut -c1-130)"
done
