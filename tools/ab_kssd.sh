#!/bin/bash
# A/B of KSSD sketch kernel builds on one box: the built library and every _variants/lib_<name>.so given.
# Usage: tools/ab_kssd.sh [variant ...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for v in base "$@" base; do
  if [ $v = base ]; then unset RTC_HIP_LIB; else export RTC_HIP_LIB=$R/_variants/lib_$v.so; fi
  echo "== $v: $(python $R/tools/run_kssd.py 25000 2000000 4 2>/dev/null | tail -3 | awk '{printf "%s ", $8}')"
done
