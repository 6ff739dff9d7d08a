#!/bin/bash
# Build an experimental variant of the native library next to the product one, for same-box A/B runs
# (GECCO_CRF_LIBRARY=<path> selects it):  tools/build_variant.sh <tag> <source in gecco_amd/csrc> [-DMACRO ...]
# Only that source is recompiled with the extra flags; the other objects are the product build's.
set -eu
TAG=$1; SRC=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
L=$R/gecco_amd/lib
python -m gecco_amd.build > /dev/null
BASE=${SRC%.*}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -x hip "$@" -c $R/gecco_amd/csrc/$SRC -o $L/${BASE}_$TAG.o.var
OBJS=$(ls $L/*.o | grep -v "/$BASE.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libgecco_crf_$TAG.so $OBJS $L/${BASE}_$TAG.o.var
rm -f $L/${BASE}_$TAG.o.var
echo $L/libgecco_crf_$TAG.so
