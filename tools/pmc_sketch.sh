#!/bin/bash
# PMC passes over the sketch kernel only (tools/run_sketch.py): instruction counts and busy cycles.
# Usage (on the GPU box): bash tools/pmc_sketch.sh <outdir> [genomes] [length]
OUT=${1:-gpurun_out/pmc_sketch}; N=${2:-2000}; L=${3:-5000000}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/$OUT
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" \
           "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAIT_ANY" \
           "SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $R/$OUT/p$i -- python $R/tools/run_sketch.py $N $L 1 > $R/$OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob("$R/$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sketch_minhash_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
steps = $N * ($L / 64.0)
for k in sorted(tot):
    v = tot[k] / max(cnt[k], 1)
    print(f"{k:28s} per launch {v:.6g}   per wave-step {v/steps:.3f}")
PY
