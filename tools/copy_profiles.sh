#!/bin/bash
# Copies what tools/collect_profiles.sh left under gpurun_out/prof into profiles/ under the round's tag.
# (the newest run of each kind: gpurun merges every collection into the same directories)
# Usage: bash tools/copy_profiles.sh [tag]
TAG=${1:-r04}
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/gpurun_out/prof
cp $P/bench_n1.jsonl $R/profiles/${TAG}_bench_n1.jsonl
cp $(ls -t $(find $P/stats -name "*kernel_stats.csv") | head -1) $R/profiles/${TAG}_bench_kernel_stats.csv
cp $P/${TAG}_pmc_traffic.json $R/profiles/${TAG}_pmc_traffic.json
cp $P/kssd/bench_kssd_n1.jsonl $R/profiles/${TAG}_kssd_bench_n1.jsonl
cp $(ls -t $(find $P/kssd/stats -name "*kernel_stats.csv") | head -1) $R/profiles/${TAG}_kssd_kernel_stats.csv
cp $P/${TAG}_kssd_pmc_traffic.json $R/profiles/${TAG}_kssd_pmc_traffic.json
cp $P/kssd_packed/bench_kssd_packed_n1.jsonl $R/profiles/${TAG}_kssd_packed_bench_n1.jsonl
cp $(ls -t $(find $P/kssd_packed/stats -name "*kernel_stats.csv") | head -1) $R/profiles/${TAG}_kssd_packed_kernel_stats.csv
cp $P/${TAG}_kssd_packed_pmc_traffic.json $R/profiles/${TAG}_kssd_packed_pmc_traffic.json
cp $(ls -t $(find $P/greedy_stats -name "*kernel_stats.csv") | head -1) $R/profiles/${TAG}_greedy_kernel_stats.csv
grep -v "^[EWI]2026\|rocprofv3\|amdgpu.ids" $P/greedy.log > $R/profiles/${TAG}_greedy_run.log
cp $P/${TAG}_greedy_pmc_traffic.json $R/profiles/${TAG}_greedy_pmc_traffic.json
cp $P/dense/bench_dense.jsonl $R/profiles/${TAG}_dense_bench.jsonl
cp $(ls -t $(find $P/dense/stats -name "*kernel_stats.csv") | head -1) $R/profiles/${TAG}_dense_kernel_stats.csv
cp $P/${TAG}_dense_pmc_traffic.json $R/profiles/${TAG}_dense_pmc_traffic.json
ls -la $R/profiles/${TAG}_*
