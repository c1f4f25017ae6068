#!/bin/bash
# Stall / instruction-cache / LDS-queue PMC passes over one kernel of a driver script.
# Usage (on the GPU box): bash tools/pmc_stall.sh <outdir> <kernel-name-substring> <units per launch / 64> -- <driver> [args]
OUT=$1; KERN=$2; STEPS=$3; shift 4
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/$OUT
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INSTS_LDS" \
           "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $R/$OUT/p$i -- python $R/"$@" > $R/$OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); cnt = collections.defaultdict(set)
for f in glob.glob("$R/$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$KERN" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]].add(r["Dispatch_Id"])
steps = float($STEPS)
v = {k: tot[k] / max(len(cnt[k]), 1) for k in tot}
for k in sorted(v):
    print(f"{k:28s} per launch {v[k]:.6g}   per wave-step {v[k]/steps:.3f}")
PY
