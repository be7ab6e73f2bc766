set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4i
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/pmc -- python $R/tools/perf_config3.py 2000 1000 > $O/pmc.log 2>&1 || { tail -30 $O/pmc.log; exit 1; }
python $R/tools/pmc_summary.py $O/pmc | grep -A4 "k_fpc64_decode"
rm -rf $O/pmc
