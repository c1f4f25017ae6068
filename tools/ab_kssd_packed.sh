#!/bin/bash
# A/B of builds of the packed KSSD sketch kernel on one box: the built library, then every _variants/lib_<name>.so given,
# then the built library again.  Prints the sketch-phase milliseconds of tools/run_kssd_packed.py (ascii / unpack+ascii / packed).
# Usage: tools/ab_kssd_packed.sh [variant ...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for v in base "$@" base; do
  if [ $v = base ]; then unset RTC_HIP_LIB; else export RTC_HIP_LIB=$R/_variants/lib_$v.so; fi
  echo "== $v: $(python $R/tools/run_kssd_packed.py --genomes ${AB_GENOMES:-25000} --reps 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('identical', d['identical'], ' ascii', d['ascii']['ms'], ' packed', d['packed']['ms'])")"
done
