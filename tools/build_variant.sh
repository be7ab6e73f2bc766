#!/bin/bash
# A measurement build of the library with another k_fpc32_sweep.hip: tools/build_variant.sh NAME "-DFLAG ..."  ->  gpurun_variants/libtrico_NAME.so
# (the other objects are the ones trico_amd/build already holds; select with TRICO_AMD_LIB)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift
mkdir -p $R/gpurun_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I$R/include -I$R/trico_amd/csrc/hip $@ -c $R/trico_amd/csrc/hip/k_fpc32_sweep.hip -o /tmp/sweep_$N.o
OBJS=$(ls $R/trico_amd/build/*.o | grep -v "hooks.o" | grep -v k_fpc32_sweep.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o $R/gpurun_variants/libtrico_$N.so $OBJS /tmp/sweep_$N.o -ldl
echo built gpurun_variants/libtrico_$N.so
