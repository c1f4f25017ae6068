#!/bin/bash
# Same-box A/B of the pair phase: the in-tree library against every _variants/lib_<name>.so (tools/build_variant.sh) -- the headline twice
# (bench.py --no-extra: pair_ms), then BASELINE configs 5 and 3 on one GPU (--only config5_1gpu / config3_1gpu).
# Usage (on the GPU box): bash tools/ab_join_sweep.sh <outdir under gpurun_out>
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in base $(ls _variants | sed 's/lib_//; s/.so//'); do
  if [ $v = base ]; then unset RTC_HIP_LIB; else export RTC_HIP_LIB=_variants/lib_$v.so; fi
  python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline 2>/dev/null | grep -o "pair_ms\": [0-9.]*" | sed "s/^/$v headline /" >> $O/sweep.txt
done; done
for v in base $(ls _variants | sed 's/lib_//; s/.so//'); do
  if [ $v = base ]; then unset RTC_HIP_LIB; else export RTC_HIP_LIB=_variants/lib_$v.so; fi
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --only config5_1gpu 2>/dev/null | grep -o "pair_ms\": [0-9.]*" | sed "s/^/$v config5 /" >> $O/sweep.txt
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --only config3_1gpu 2>/dev/null | grep -o "pair_ms\": [0-9.]*" | sed "s/^/$v config3 /" >> $O/sweep.txt
done
cat $O/sweep.txt
