#!/bin/bash
# Instruction and cycle counters of ONE kernel for the built library and for variants (one rocprofv3 --pmc pass each).
# Usage: tools/pmc_ab.sh <kernel-name-substring> "<driver and args>" [variant ...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
KERN=$1; DRV=$2; shift 2
cd /tmp && export TMPDIR=/tmp
for v in base "$@"; do
  if [ $v = base ]; then unset RTC_HIP_LIB; else export RTC_HIP_LIB=$R/_variants/lib_$v.so; fi
  rm -rf /tmp/pmcab_$v
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmcab_$v -- python $R/$DRV > /tmp/pmcab_$v.log 2>&1
  python - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); cnt = collections.defaultdict(set)
for f in glob.glob("/tmp/pmcab_$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$KERN" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]].add(r["Dispatch_Id"])
print("== $v:", {k: round(tot[k] / max(len(cnt[k]), 1) / 1e6, 2) for k in sorted(tot)}, "(millions per launch)")
PY
done
