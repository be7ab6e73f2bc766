set -e
R=$GRAFT_REPO_ROOT
cd $R
echo "## kept"; timeout -k 10 200 python tools/perf_float_decode_nocheck.py 2>&1 | tail -3
echo "## no wait in F values before a D value (unsafe)"
TRICO_AMD_LIB=$R/tools/_exp/libtrico_no_fd_wait.so timeout -k 10 200 python tools/perf_float_decode_nocheck.py 2>&1 | tail -3
