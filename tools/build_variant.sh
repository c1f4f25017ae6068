#!/bin/bash
# Builds _variants/lib_<name>.so: the library with ONE translation unit recompiled with extra flags (A/B timing runs,
# selected at run time with RTC_HIP_LIB).  Usage: tools/build_variant.sh <name> <file.hip> [-DFLAG ...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; SRC=$2; shift 2
C=$R/rabbittclust_amd/csrc
mkdir -p $R/_variants /tmp/variant_$NAME
make -C $C -j8 -s
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -ffp-contract=off "$@" -c $C/$SRC -o /tmp/variant_$NAME/obj.o
OBJS=$(ls $C/_build/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/_variants/lib_$NAME.so $OBJS /tmp/variant_$NAME/obj.o -ldl
echo "built _variants/lib_$NAME.so"
