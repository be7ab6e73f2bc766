# SQ counters of the float encoder's sweep in five passes: bash tools/gpu_sq_passes.sh <variant lib name> <mesh>...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/variants; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
V=$1; shift
export TRICO_AMD_LIB=$R/gpurun_variants/libtrico_$V.so
for m in "$@"; do
  echo "## $V $m"
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_INSTS_SMEM" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES" \
             "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_IFETCH SQ_BUSY_CYCLES" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL" \
             "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_LDS"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$i -- python $R/tools/perf_fpc32.py $m > $O/pmc_$i.log 2>&1
    python $R/tools/pmc_summary.py $O/pmc_$i | grep -A7 "k_fpc32_sweep" | grep -v "^k_"
    rm -rf $O/pmc_$i
  done
done
