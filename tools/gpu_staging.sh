#!/bin/bash
# Host-pointer path (staging.hip): its GPU tests, then config 2 through the plain API with the runtime's staging (0) and the ring.
set -e
mkdir -p gpurun_out/staging
timeout -k 10 600 python -m pytest tests/test_gpu_staging.py tests/test_gpu_cli.py -m gpu -x -q --timeout=250 2>&1 | tee gpurun_out/staging/pytest.log
nproc
for T in 0 default; do
  echo "## TRICO_HIP_STAGE_THREADS=$T"
  if [ $T = default ]; then unset TRICO_HIP_STAGE_THREADS; else export TRICO_HIP_STAGE_THREADS=$T; fi
  timeout -k 10 200 python tools/perf_host_pointers.py 2>&1 | tee gpurun_out/staging/threads_$T.log
done
