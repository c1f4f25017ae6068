#!/bin/bash
# The pair phase of bench.py over family sizes (FAMS="10 40 ..."), by the cost rule (RTC_PAIR_JOIN=1), tiled kernel only (0), join forced (2):
# which path the rule picks against what either path costs.  Usage (on the GPU box): bash tools/ab_join_families.sh <outdir under gpurun_out>
cd $GRAFT_REPO_ROOT; O=gpurun_out/$1; mkdir -p $O
for fam in ${FAMS:-10 16 20 25 40}; do
 for pj in 1 0 2; do
  RTC_PAIR_JOIN=$pj python bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline --family $fam 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); p=d['phase_ms']; print('family $fam RTC_PAIR_JOIN=$pj pair_ms %.3f mst_ms %.3f path %d edges %d' % (p['pair_ms'], p['mst_ms'], p['pair_path'], p['cand_edges']))" >> $O/fam.txt
 done
done
cat $O/fam.txt
