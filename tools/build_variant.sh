#!/bin/bash
# Build an experimental variant of the native library next to the product one, for same-box A/B runs
# (GECCO_CRF_LIBRARY=<path> selects it):  tools/build_variant.sh <tag> [-DMACRO ...]
# Only crf_kernels.hip is recompiled with the extra flags; the other objects are the product build's.
set -eu
TAG=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
L=$R/gecco_amd/lib
python -m gecco_amd.build > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -x hip "$@" -c $R/gecco_amd/csrc/crf_kernels.hip -o $L/crf_kernels_$TAG.o
OBJS=$(ls $L/*.o | grep -v "crf_kernels")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libgecco_crf_$TAG.so $OBJS $L/crf_kernels_$TAG.o
rm -f $L/crf_kernels_$TAG.o
echo $L/libgecco_crf_$TAG.so
