#!/bin/bash
# L2 / fabric-side counters of the sketch kernel (which request sizes does FETCH_SIZE tally here?)
OUT=${1:-gpurun_out/pmc_tcc}; N=${2:-10000}; L=${3:-5000000}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/$OUT
rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > $R/$OUT/avail.txt
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum" "FETCH_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $R/$OUT/p$i -- python $R/tools/run_sketch.py $N $L 1 > $R/$OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); cnt = collections.defaultdict(set)
for f in glob.glob("$R/$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sketch_minhash_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(tot):
    print(f"{k:28s} per launch {tot[k]/max(len(cnt[k]),1):.6g}")
print("algorithmic bytes", $N*$L)
PY
