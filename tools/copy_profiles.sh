#!/bin/bash
# Copies what tools/collect_profiles.sh left under gpurun_out/prof into profiles/ under the round's tag (the newest run of each
# kind: gpurun merges every collection into the same directories), then rewrites profiles/README.md's table from the CSVs.
# Usage: bash tools/copy_profiles.sh [tag]
TAG=${1:-r06}
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/gpurun_out/prof
for S in minhash minhash_ascii minhash_packed kssd kssd_packed greedy dense; do
  [ -d $P/$S ] || continue
  N=${TAG}_$S; [ $S = minhash ] && N=${TAG}
  [ -s $P/$S/bench.jsonl ] && cp $P/$S/bench.jsonl $R/profiles/${N}_bench.jsonl
  CSV=$(ls -t $(find $P/$S/stats -name "*kernel_stats.csv" 2>/dev/null) 2>/dev/null | head -1)
  [ -n "$CSV" ] && cp $CSV $R/profiles/${N}_kernel_stats.csv
  [ -s $P/${N}_pmc_traffic.json ] && cp $P/${N}_pmc_traffic.json $R/profiles/${N}_pmc_traffic.json
done
ls -la $R/profiles/${TAG}_*
python $R/tools/profiles_readme.py $TAG > /dev/null
