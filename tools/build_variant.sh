#!/bin/bash
# A measurement build of the library with another build of ONE kernel source: tools/build_variant.sh NAME "-DFLAG ..." [source.hip]  ->  gpurun_variants/libtrico_NAME.so
# (default source: k_fpc32_sweep.hip; the other objects are the ones trico_amd/build already holds; select the library with TRICO_AMD_LIB)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1
F=$2
S=${3:-k_fpc32_sweep.hip}
mkdir -p $R/gpurun_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I$R/include -I$R/trico_amd/csrc/hip $F -c $R/trico_amd/csrc/hip/$S -o /tmp/variant_$N.o
OBJS=$(ls $R/trico_amd/build/*.o | grep -v "hooks.o" | grep -v "$S.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o $R/gpurun_variants/libtrico_$N.so $OBJS /tmp/variant_$N.o -ldl
echo built gpurun_variants/libtrico_$N.so
