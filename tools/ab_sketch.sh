#!/bin/bash
# Same-box A/B of the MinHash sketch kernel: the in-tree library against _variants/lib_<name>.so (tools/build_variant.sh),
# alternating, over a list of shapes "n length reps size".  Usage (on the GPU box): bash tools/ab_sketch.sh <name> "<shape>" ...
cd $GRAFT_REPO_ROOT
NAME=$1; shift
for shape in "$@"; do
  for rep in 1 2; do
    for v in base $NAME; do
      if [ $v = base ]; then unset RTC_HIP_LIB; else export RTC_HIP_LIB=$GRAFT_REPO_ROOT/_variants/lib_$v.so; fi
      echo -n "$v: "; python tools/run_sketch.py $shape 2>/dev/null | tail -1
    done
  done
done
