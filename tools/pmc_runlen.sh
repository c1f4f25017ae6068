#!/bin/bash
# time + fabric read requests of the sketch kernel for library variants (lane run length experiments)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" != cur ]; then export RTC_HIP_LIB=$R/_variants/lib_$v.so; else unset RTC_HIP_LIB; fi
  t=$(python $R/tools/run_sketch.py 10000 5000000 2 2>&1 | tail -1)
  rm -rf /tmp/pr_$v; timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum --output-format csv -d /tmp/pr_$v -- python $R/tools/run_sketch.py 10000 5000000 1 > /dev/null 2>&1
  b=$(python - <<PY
import csv, glob
n128 = n64 = 0
for f in glob.glob("/tmp/pr_$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sketch_minhash_kernel" in r["Kernel_Name"]:
            if r["Counter_Name"].startswith("TCC_EA0_RDREQ_128B"): n128 += float(r["Counter_Value"])
            if r["Counter_Name"].startswith("TCC_EA0_RDREQ_64B"): n64 += float(r["Counter_Value"])
print(f"fabric reads {(n128*128+n64*64)/1e9:.1f} GB = {(n128*128+n64*64)/50e9:.2f} x algorithmic")
PY
)
  echo "$v: $t | $b"
done
