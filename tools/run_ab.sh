# A/B of library builds on the float encoder; usage: gpurun -- bash tools/run_ab.sh "v0 v2" "grid walk" [ENV=VALUE]   (trico_amd/lib/ab_<name>.so)
cd /tmp
R=$GRAFT_REPO_ROOT
for rep in 1 2; do for kind in $2; do for v in $1; do
  echo "== $kind $v"
  env $3 TRICO_AMD_LIB=$R/trico_amd/lib/ab_$v.so timeout -k 10 120 python $R/tools/perf_fpc32.py $kind | grep "kernel span"
done; done; done
