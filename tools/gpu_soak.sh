# soak runs of a round's end: bash tools/gpu_soak.sh  (-> gpurun_out/soak_*.log)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
SOAK_DEVICE=1 timeout -k 10 500 python tools/soak.py 500 61 > $O/soak_device.log 2>&1; tail -2 $O/soak_device.log
SOAK_DEVICE=1 SOAK_BIG=1 SOAK_MAXN=20000000 timeout -k 10 400 python tools/soak.py 60 62 > $O/soak_device_big.log 2>&1; tail -2 $O/soak_device_big.log
TRICO_HIP_ENCODE_VERIFY=1 timeout -k 10 400 python tools/soak.py 400 63 > $O/soak_verify.log 2>&1; tail -2 $O/soak_verify.log
TRICO_HIP_ENCODE_VERIFY=1 SOAK_BIG=1 SOAK_MAXN=20000000 timeout -k 10 400 python tools/soak.py 50 64 > $O/soak_verify_big.log 2>&1; tail -2 $O/soak_verify_big.log
timeout -k 10 300 python tools/stress_concurrent.py > $O/stress_r06b.log 2>&1; tail -3 $O/stress_r06b.log
