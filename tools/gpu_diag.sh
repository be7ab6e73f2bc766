# per-wave timelines of the sweep (diagnostic build gpurun_variants/libtrico_diag6.so, tools/build_variant.sh diag6 "-DTRICO_SWEEP_DIAG -DTRICO_HIP_TEST_HOOKS"):
# bash tools/gpu_diag.sh BETA...   (TRICO_FPC32_STAGGER per run)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/diag6
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export TRICO_AMD_LIB=$R/gpurun_variants/libtrico_diag6.so
for b in "$@"; do
for m in grid walk; do
  TRICO_FPC32_STAGGER=$b TRICO_SWEEP_DIAG_FILE=$O/diag_${m}_$b.bin timeout -k 10 200 python $R/tools/perf_fpc32.py $m > $O/${m}_$b.log 2>&1 && \
  { echo "## $m beta $b"; grep "kernel span" $O/${m}_$b.log; python $R/tools/diag_analyze.py $O/diag_${m}_$b.bin | grep -v "XCC\|segments"; } > $O/analysis_${m}_$b.txt 2>&1
  cat $O/analysis_${m}_$b.txt
done
done
