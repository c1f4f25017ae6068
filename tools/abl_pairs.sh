#!/bin/bash
# Kernel time of pair_tiled_kernel for the built library and for every _variants/lib_<name>.so given (A/B and ablation
# runs on one box).  Usage: tools/abl_pairs.sh "<driver args, e.g. tools/run_pairs.py minhash 10000 4>" [variant ...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ARGS=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in base "$@"; do
  if [ $v = base ]; then unset RTC_HIP_LIB; else export RTC_HIP_LIB=$R/_variants/lib_$v.so; fi
  rm -rf /tmp/abl_$v
  RTC_PAIR_JOIN=${RTC_PAIR_JOIN:-0} timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_$v -- python $R/$ARGS > /tmp/abl_$v.log 2>&1
  echo "== $v: $(grep -c 'pair phase' /tmp/abl_$v.log) runs, last: $(grep 'pair phase' /tmp/abl_$v.log | tail -1)"
  python $R/tools/kstats.py /tmp/abl_$v | grep -E "pair_tiled|transpose|slice_|plan_stats" | head -6
done
