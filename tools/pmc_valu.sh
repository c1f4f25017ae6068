#!/bin/bash
# one PMC pass (instruction counts) over one kernel: bash tools/pmc_valu.sh <kernel-substring> <bases per launch> -- <driver> [args]
KERN=$1; BASES=$2; shift 3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pv && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d /tmp/pv -- python $R/"$@" > /tmp/pv.log 2>&1
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); cnt = collections.defaultdict(set)
for f in glob.glob("/tmp/pv/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$KERN" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(tot):
    v = tot[k] / max(len(cnt[k]), 1)
    print(f"{k:16s} per launch {v:.5g}  per 64 bases {v / ($BASES / 64.0):.3f}")
PY
