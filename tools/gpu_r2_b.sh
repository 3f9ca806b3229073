#!/bin/bash
# Round 2, visit B: timing-only ablation of the dominant fused-pair kernel + kernel stats of the fused AMP pair (C3).
OUT=gpurun_out/r2_b
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
for round in 1 2; do
  for m in 0 2 4 8 16 32 64 96 128 256 6 14 30; do timeout 60 tests/experiments/pair_ablate_$m 5; done
done > $OUT/pair_ablate.txt 2>&1
cat $OUT/pair_ablate.txt | sort | head -60
( cd /tmp && AMP_FUSE_AMP=1 timeout 120 rocprofv3 --output-format csv --kernel-trace --stats -d $REPO/$OUT/c3f -o kt -- python $REPO/tools/bench_configs.py --only c3 --reps 3 > $REPO/$OUT/c3f.json 2> $REPO/$OUT/c3f.err )
head -14 $OUT/c3f/kt_kernel_stats.csv | cut -c1-200
rm -f $OUT/c3f/kt_kernel_trace.csv
find $OUT -name "*.db" -delete 2>/dev/null; find $OUT -name "*agent_info*" -delete; du -sh $OUT
