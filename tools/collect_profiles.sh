#!/bin/bash
# Collects the round's judged profile artifacts on the GPU box (run through gpurun).  Per section:
#   bench.py JSON line                                     -> gpurun_out/prof/<section>/bench.jsonl
#   rocprofv3 --kernel-trace --stats of the same command   -> gpurun_out/prof/<section>/stats/  (kernel_stats.csv)
#   rocprofv3 --pmc passes (own runs, --pmc only)          -> gpurun_out/prof/<section>/pmc*/  and <tag>_<section>_pmc_traffic.json
# Sections: minhash (BASELINE config[1] from the 2-bit staging format, the judged line with every extra), minhash_ascii (the same
# batch resident as characters), kssd / kssd_packed (config[4]'s per-GPU shape), greedy (config[3]), dense (tiled N x N kernel regime).
# Usage: bash tools/collect_profiles.sh [tag] [section ...]     (tag names the files, e.g. r05; no section = all)
TAG=${1:-r06}; shift
SECTIONS=${@:-minhash minhash_ascii kssd kssd_packed greedy dense}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PMCGROUPS=("FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
        "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
        "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE")
# section <dir> <bench flags of the JSON line> -- <flags of the stats run> -- <flags of the PMC runs>
run_section() {
  local D=$OUT/$1; shift
  mkdir -p $D
  local full=() stats=() pmc=() which=full
  for a in "$@"; do
    if [ "$a" = "--" ]; then if [ $which = full ]; then which=stats; else which=pmc; fi; continue; fi
    case $which in full) full+=("$a");; stats) stats+=("$a");; pmc) pmc+=("$a");; esac
  done
  python $R/bench.py "${full[@]}" > $D/bench.jsonl 2> $D/bench.err
  tail -n 1 $D/bench.jsonl | tail -c 600; echo
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python $R/bench.py "${stats[@]}" > $D/stats.log 2>&1
  python $R/tools/kstats.py $D/stats | head -10
  local i=0
  for grp in "${PMCGROUPS[@]}"; do
    i=$((i+1))
    timeout 900 rocprofv3 --pmc $grp --output-format csv -d $D/pmc$i -- python $R/bench.py "${pmc[@]}" > $D/pmc$i.log 2>&1
  done
}
for S in $SECTIONS; do
  case $S in
    minhash)   # the judged line: BASELINE config[1], batch resident in the 2-bit staging format, every extra
      run_section minhash --steps 5 --warmup 2 -- --steps 3 --warmup 1 --no-cpu-baseline --no-extra -- --steps 1 --warmup 0 --no-cpu-baseline --no-extra
      python $R/tools/make_pmc_json.py $OUT/minhash $TAG minhash 10000 5000000 packed > $OUT/${TAG}_pmc_traffic.json;;
    minhash_ascii)   # the same batch resident as characters
      run_section minhash_ascii --staging ascii --steps 5 --warmup 2 --no-extra -- --staging ascii --steps 3 --warmup 1 --no-cpu-baseline --no-extra -- --staging ascii --steps 1 --warmup 0 --no-cpu-baseline --no-extra
      python $R/tools/make_pmc_json.py $OUT/minhash_ascii ${TAG}_minhash_ascii minhash 10000 5000000 > $OUT/${TAG}_minhash_ascii_pmc_traffic.json;;
    kssd)
      run_section kssd --mode kssd --staging ascii --steps 3 --warmup 1 -- --mode kssd --staging ascii --steps 3 --warmup 1 --no-cpu-baseline -- --mode kssd --staging ascii --steps 1 --warmup 0 --no-cpu-baseline
      python $R/tools/make_pmc_json.py $OUT/kssd ${TAG}_kssd kssd 25000 2000000 > $OUT/${TAG}_kssd_pmc_traffic.json;;
    kssd_packed)
      run_section kssd_packed --mode kssd --steps 3 --warmup 1 -- --mode kssd --steps 3 --warmup 1 --no-cpu-baseline -- --mode kssd --steps 1 --warmup 0 --no-cpu-baseline
      python $R/tools/make_pmc_json.py $OUT/kssd_packed ${TAG}_kssd_packed kssd 25000 2000000 packed > $OUT/${TAG}_kssd_packed_pmc_traffic.json;;
    greedy)
      # 50 000 prefix genomes of 0.4 .. 2 Mbp: 64 Gbp, mean length 1.28 Mbp (the "derived" per-step figures use genomes x this length)
      run_section greedy --only greedy -- --only greedy -- --only greedy --extra-steps 1
      python $R/tools/make_pmc_json.py $OUT/greedy ${TAG}_greedy greedy 50000 1280000 > $OUT/${TAG}_greedy_pmc_traffic.json;;
    dense)
      run_section dense --only dense_pairs -- --only dense_pairs -- --only dense_pairs --extra-steps 1
      python $R/tools/make_pmc_json.py $OUT/dense ${TAG}_dense dense_pairs 10000 500000 > $OUT/${TAG}_dense_pmc_traffic.json;;
  esac
done
# what travels back is the summaries: the raw counter and trace CSVs (tens of MB per run) stay on the box
find $OUT -type d -name "pmc[0-9]*" -prune -exec rm -rf {} +
find $OUT -path "*stats*" -type f ! -name "*kernel_stats.csv" ! -name "stats.log" -delete
python - <<PY
import glob, json
for f in sorted(glob.glob("$OUT/${TAG}_*pmc_traffic.json")):
    d = json.load(open(f))
    for k, v in d["kernels"].items():
        print(f.split("/")[-1], k, {a: b for a, b in v.items() if a in ("hbm_bytes_per_launch", "derived")})
PY
