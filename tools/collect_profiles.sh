#!/bin/bash
# Collects the round's judged profile artifacts on the GPU box (run through gpurun):
#   1. bench.py JSON lines (minhash N=1 default; --mode kssd [--staging packed])       -> gpurun_out/prof/bench_*.jsonl
#   2. rocprofv3 --kernel-trace --stats of the same commands            -> gpurun_out/prof/stats*/  (kernel_stats.csv)
#   3. rocprofv3 --pmc passes (own runs, --pmc only)                     -> gpurun_out/prof/pmc*/  and *_pmc_traffic.json
#   4. greedy (BASELINE config 4, 50 000 containment sketches) kernel stats + PMC passes of its sketch kernel
#   5. the dense regime of the pair phase (bench.py --only dense_pairs: 10 families of 1 000, tiled N x N kernel): stats + PMC
# Usage: bash tools/collect_profiles.sh [tag]     (tag names the files, e.g. r02)
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PMCGROUPS=("FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
        "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
        "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE")
# ---- MinHash, BASELINE config[1] ----
python $R/bench.py --steps 5 --warmup 2 > $OUT/bench_n1.jsonl 2> $OUT/bench_n1.err
tail -c 600 $OUT/bench_n1.jsonl; echo
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/stats.log 2>&1
python $R/tools/kstats.py $OUT/stats | head -12
i=0
for grp in "${PMCGROUPS[@]}"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc$i -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra > $OUT/pmc$i.log 2>&1
done
python $R/tools/make_pmc_json.py $OUT $TAG > $OUT/${TAG}_pmc_traffic.json
# ---- KSSD (--fast), BASELINE config[4] per-GPU shape ----
K=$OUT/kssd; mkdir -p $K
python $R/bench.py --mode kssd --steps 3 --warmup 1 > $K/bench_kssd_n1.jsonl 2> $K/bench_kssd_n1.err
tail -c 400 $K/bench_kssd_n1.jsonl; echo
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $K/stats -- python $R/bench.py --mode kssd --steps 3 --warmup 1 --no-cpu-baseline > $K/stats.log 2>&1
python $R/tools/kstats.py $K/stats | head -8
i=0
for grp in "${PMCGROUPS[@]}"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d $K/pmc$i -- python $R/bench.py --mode kssd --steps 1 --warmup 0 --no-cpu-baseline > $K/pmc$i.log 2>&1
done
python $R/tools/make_pmc_json.py $K ${TAG}_kssd kssd 25000 2000000 > $OUT/${TAG}_kssd_pmc_traffic.json
# ---- KSSD from the 2-bit staging format (rtc_sketch_kssd_packed_dev), same shape ----
KP=$OUT/kssd_packed; mkdir -p $KP
python $R/bench.py --mode kssd --staging packed --steps 3 --warmup 1 > $KP/bench_kssd_packed_n1.jsonl 2> $KP/bench_kssd_packed_n1.err
tail -c 400 $KP/bench_kssd_packed_n1.jsonl; echo
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $KP/stats -- python $R/bench.py --mode kssd --staging packed --steps 3 --warmup 1 --no-cpu-baseline > $KP/stats.log 2>&1
python $R/tools/kstats.py $KP/stats | head -8
i=0
for grp in "${PMCGROUPS[@]}"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d $KP/pmc$i -- python $R/bench.py --mode kssd --staging packed --steps 1 --warmup 0 --no-cpu-baseline > $KP/pmc$i.log 2>&1
done
python $R/tools/make_pmc_json.py $KP ${TAG}_kssd_packed kssd 25000 2000000 packed > $OUT/${TAG}_kssd_packed_pmc_traffic.json
# ---- greedy, BASELINE config[3] ----
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/greedy_stats -- python $R/tools/run_configs.py greedy 50000 2000000 > $OUT/greedy.log 2>&1
tail -3 $OUT/greedy.log
python $R/tools/kstats.py $OUT/greedy_stats | head -8
i=0
for grp in "${PMCGROUPS[@]}"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d $OUT/greedy_pmc/pmc$i -- python $R/bench.py --only greedy --extra-steps 1 > $OUT/greedy_pmc$i.log 2>&1
done
# 50 000 prefix genomes of 0.4 .. 2 Mbp: 64 Gbp, mean length 1.28 Mbp (the "derived" per-step figures use genomes x this length)
python $R/tools/make_pmc_json.py $OUT/greedy_pmc ${TAG}_greedy greedy 50000 1280000 > $OUT/${TAG}_greedy_pmc_traffic.json
# ---- dense regime of the pair phase (tiled N x N kernel) ----
D=$OUT/dense; mkdir -p $D
python $R/bench.py --only dense_pairs > $D/bench_dense.jsonl 2> $D/bench_dense.err
tail -c 600 $D/bench_dense.jsonl; echo
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python $R/bench.py --only dense_pairs > $D/stats.log 2>&1
python $R/tools/kstats.py $D/stats | head -8
i=0
for grp in "${PMCGROUPS[@]}"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d $D/pmc$i -- python $R/bench.py --only dense_pairs --extra-steps 1 > $D/pmc$i.log 2>&1
done
python $R/tools/make_pmc_json.py $D ${TAG}_dense dense_pairs 10000 500000 > $OUT/${TAG}_dense_pmc_traffic.json
python - <<PY
import json
for f in ("$OUT/${TAG}_pmc_traffic.json", "$OUT/${TAG}_kssd_pmc_traffic.json", "$OUT/${TAG}_kssd_packed_pmc_traffic.json", "$OUT/${TAG}_greedy_pmc_traffic.json", "$OUT/${TAG}_dense_pmc_traffic.json"):
    d = json.load(open(f))
    for k, v in d["kernels"].items():
        print(k, {a: b for a, b in v.items() if a in ("hbm_bytes_per_launch", "derived")})
PY
