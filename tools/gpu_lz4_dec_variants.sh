# LZ4 decode variants (gpurun_variants/libtrico_NAME.so, tools/build_variant.sh NAME "-D..." k_lz4_pdecode.hip): decode time per mesh.  bash tools/gpu_lz4_dec_variants.sh NAME...
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/lz4_dec_variants
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  export TRICO_AMD_LIB=$R/gpurun_variants/libtrico_$v.so
  for m in grid walk; do
    timeout -k 10 200 python $R/tools/perf_lz4.py $m > $O/${v}_$m.log 2>&1
    echo "## $v $m: $(grep 'decode iter' $O/${v}_$m.log | tr '\n' ' ')"
  done
done
