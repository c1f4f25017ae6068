#!/bin/bash
# PMC passes over one kernel of a driver script: instruction counts, busy cycles, fabric traffic.
# Usage (on the GPU box): bash tools/pmc_kernel.sh <outdir> <kernel-name-substring> <units per launch / 64> -- <driver> [args]
OUT=$1; KERN=$2; STEPS=$3; shift 4
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/$OUT
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR" \
           "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
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
rd = 128*v.get("TCC_EA0_RDREQ_128B_sum",0) + 64*v.get("TCC_EA0_RDREQ_64B_sum",0) + 32*v.get("TCC_EA0_RDREQ_32B_sum",0)
w64 = v.get("TCC_EA0_WRREQ_64B_sum",0); wr = 64*w64 + 32*max(v.get("TCC_EA0_WRREQ_sum",0)-w64,0)
print(f"fabric bytes per launch: read {rd:.4g} write {wr:.4g} total {rd+wr:.4g}")
PY
