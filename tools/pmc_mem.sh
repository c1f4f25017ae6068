#!/bin/bash
# PMC passes for the memory side of one kernel: L1 (TCP) / L2 (TCC) requests and stalls.
# Usage (on the GPU box): bash tools/pmc_mem.sh <outdir> <kernel-name-substring> -- <driver> [args]
OUT=$1; KERN=$2; shift 3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/$OUT
i=0
for grp in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum TD_TD_BUSY_sum" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $R/$OUT/m$i -- python $R/"$@" > $R/$OUT/m$i.log 2>&1 || tail -3 $R/$OUT/m$i.log
done
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); cnt = collections.defaultdict(set)
for f in glob.glob("$R/$OUT/m*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$KERN" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]].add(r["Dispatch_Id"])
v = {k: tot[k] / max(len(cnt[k]), 1) for k in tot}
for k in sorted(v):
    print(f"{k:36s} per launch {v[k]:.6g}")
PY
