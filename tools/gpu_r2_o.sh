#!/bin/bash
# Round 2, visit O: per-kernel comparison of the per-tile pair kernel and the 2 x 2-blocked variant for every C = 128 pair
OUT=gpurun_out/r2_o
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
for v in 0 7 6; do
  ( cd /tmp && AMP_STRIP_C128=$v timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $REPO/$OUT/p$v -o kt -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $REPO/$OUT/bench_$v.json 2> $REPO/$OUT/p$v.err )
  echo "== AMP_STRIP_C128=$v"; grep "pair_f16x3_kernel<[0-9]*, 4, 1, 3\|pair_strip_kernel" $OUT/p$v/kt_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
  rm -f $OUT/p$v/kt_kernel_trace.csv
done
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete
