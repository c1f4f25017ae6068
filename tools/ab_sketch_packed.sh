#!/bin/bash
# Same-box A/B of the packed-input MinHash sketch kernel: the in-tree library against _variants/lib_<name>.so (tools/build_variant.sh),
# alternating, over shapes "n length reps size k".  Usage (on the GPU box): bash tools/ab_sketch_packed.sh "<name> <name> ..." "<shape>" ...
cd $GRAFT_REPO_ROOT
NAMES=$1; shift
for shape in "$@"; do
  for rep in 1 2; do
    for v in base $NAMES; do
      if [ $v = base ]; then unset RTC_HIP_LIB; else export RTC_HIP_LIB=$GRAFT_REPO_ROOT/_variants/lib_$v.so; fi
      echo -n "$v: "; python tools/run_sketch_packed.py $shape packed 2>/dev/null | tail -2 | tr '\n' ' '; echo
    done
  done
done
