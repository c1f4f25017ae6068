#!/bin/bash
# Collects the round's judged profile artifacts on the GPU box (run through gpurun):
#   1. bench.py JSON line                          -> gpurun_out/prof/bench_n1.jsonl
#   2. rocprofv3 --kernel-trace --stats of bench.py -> gpurun_out/prof/stats/  (kernel_stats.csv)
#   3. rocprofv3 --pmc passes (own runs, --pmc only) -> gpurun_out/prof/pmc*/  and pmc_traffic.json
# Usage: bash tools/collect_profiles.sh [tag]     (tag names the JSON, e.g. r01)
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 3 --warmup 1 > $OUT/bench_n1.jsonl 2> $OUT/bench_n1.err
tail -c 400 $OUT/bench_n1.jsonl; echo
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/stats.log 2>&1
python $R/tools/kstats.py $OUT/stats | head -12
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc$i -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc$i.log 2>&1
done
python $R/tools/make_pmc_json.py $OUT $TAG > $OUT/${TAG}_pmc_traffic.json
python - <<PY
import json
d = json.load(open("$OUT/${TAG}_pmc_traffic.json"))
for k, v in d["kernels"].items():
    print(k, {a: b for a, b in v.items() if a in ("hbm_bytes_per_launch", "derived")})
PY
