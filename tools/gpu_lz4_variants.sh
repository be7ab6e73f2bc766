# LZ4 compress variants (gpurun_variants/libtrico_NAME.so, tools/build_variant.sh NAME "-D..." k_lz4_chunked.hip): encode time per mesh.  bash tools/gpu_lz4_variants.sh NAME...
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/lz4_variants
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  if [ $v = product ]; then unset TRICO_AMD_LIB; else export TRICO_AMD_LIB=$R/gpurun_variants/libtrico_$v.so; fi
  for m in walk grid; do
    timeout -k 10 200 python $R/tools/perf_lz4.py $m > $O/${v}_$m.log 2>&1
    echo "## $v $m: $(grep 'encode iter 2' $O/${v}_$m.log | cut -c1-40)"
  done
done 2>&1 | tee $O/summary.txt
