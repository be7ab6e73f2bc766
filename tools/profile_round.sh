# Collects the profiles committed under profiles/: kernel trace of bench.py and PMC passes over the float encoder.
# usage (GPU box): bash tools/profile_round.sh <tag>
set -e
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
echo "== kernel trace of bench.py"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python $R/bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err
python $R/tools/prof_summary.py $O/bench > $O/${TAG}_bench_config2_kernel_stats.txt
cp $O/bench.json $O/${TAG}_bench_config2.json
echo "== PMC: HBM traffic of the float encoder (separate passes)"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python $R/tools/perf_fpc32.py grid > $O/pmc_$c.log 2>&1
done
{ echo "# rocprofv3 --pmc FETCH_SIZE (own pass) and --pmc WRITE_SIZE (own pass), tools/perf_fpc32.py grid (50M float xyz vertices), per dispatch, unit KB"
  echo "# gfx950: FETCH_SIZE reports 1/2 of the bytes of streaming reads (MI355X_MICROARCH.md, HBM section) -> doubled in DESIGN.md and bench.py"
  python $R/tools/pmc_summary.py $O/pmc_FETCH_SIZE; python $R/tools/pmc_summary.py $O/pmc_WRITE_SIZE; } > $O/${TAG}_fpc32_encode_hbm_traffic_pmc.txt
echo "== PMC: SQ counters"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/pmc_sq1 -- python $R/tools/perf_fpc32.py grid > $O/pmc_sq1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq2 -- python $R/tools/perf_fpc32.py grid > $O/pmc_sq2.log 2>&1
{ echo "# rocprofv3 SQ counters of the float encoder sweeps (tools/perf_fpc32.py grid), per dispatch"
  python $R/tools/pmc_summary.py $O/pmc_sq1; python $R/tools/pmc_summary.py $O/pmc_sq2; } > $O/${TAG}_fpc32_encode_sq_counters.txt
grep "kernel span" $O/pmc_sq1.log || true
rm -rf $O/bench $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_sq1 $O/pmc_sq2
ls -la $O
