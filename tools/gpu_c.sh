set -e
cd $GRAFT_REPO_ROOT
for a in 0 1 2 3 4 7 8 15 16 24 28 31; do
  echo "== ablation $a"
  TRICO_FPC32_DEC=3 TRICO_FPC32_ABL=$a timeout -k 10 100 python tools/perf_fpc32_decode.py grid 2000 2000 2>&1 | grep "prof" | tail -1
done
