#!/bin/bash
# Builds _variants/lib_<name>.so: the library with ONE translation unit replaced by the given source (A/B runs on one box with
# RTC_HIP_LIB=_variants/lib_<name>.so).  Usage: tools/build_variant.sh name src.hip replaced_unit [hipcc flags...]
# (a source under _variants/src/ includes "../../rabbittclust_amd/csrc/rtc_internal.h")
set -e
R=$(cd "$(dirname "$0")/.." && pwd); NAME=$1; SRC=$2; UNIT=$3; shift 3
C=$R/rabbittclust_amd/csrc
mkdir -p /tmp/variant_$NAME
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -ffp-contract=off "$@" -c $SRC -o /tmp/variant_$NAME/obj.o
OBJS=$(ls $C/_build/*.o | grep -v "/${UNIT}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/_variants/lib_$NAME.so $OBJS /tmp/variant_$NAME/obj.o -ldl
echo built $NAME
