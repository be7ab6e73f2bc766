set -e
R=$GRAFT_REPO_ROOT
cd $R
for v in ${VARIANTS:-3 2}; do
  for kind in ${KINDS:-grid walk}; do
    echo "== variant $v $kind"
    TRICO_FPC32_DEC=$v timeout -k 10 200 python tools/perf_fpc32_decode.py $kind ${W:-4000} ${H:-2500} 2>&1 | grep "comp\|prof"
  done
done
